#!/usr/bin/env python3
"""bench.py -- shuffle proofs/sec (prove + verify) on MI355X, BASELINE.json's metric.

One step = one pass of the hot path over one batch: B independent 52-card decks (m=2, n=26, STARK curve
[REF barnett-smart-card-protocol/examples/round.rs:229-230]) each go through `shuffle_and_remask` (ElGamal
re-encryption + Bayer-Groth prover) and `verify_shuffle`, with every input already resident in HBM.
Synthetic data: random ciphertext decks [REF src/discrete_log_cards/tests.rs:187] obtained by re-encrypting one
random base deck on the GPU; masking factors uniform below 2^251; uniform permutations; random prover seeds.

  python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU.  Started from a plain shell, bench.py launches the N ranks ITSELF (it re-executes under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`); started by the driver under
torch.distributed.run it reads RANK / LOCAL_RANK / WORLD_SIZE from the environment.  Proofs are independent, so ranks
shard the batch with no data-path collective (weak scaling: B proofs per GPU per step); the shared parameters are
produced on rank 0 and broadcast once over RCCL; per-rank proof counts, verdict sums and times are all-gathered and
rank 0 prints ONE JSON line.

Workloads (`--workload`, BASELINE.json's configs):
  pairs    (default; configs 2 and 4) B independent prove + verify pairs per step
  chain32  (config 3) T card tables x 32 players: 32 DEPENDENT shuffles per table (deck_{j+1} = output_j), each table under its
           own aggregate key (keyed batches); every link proved (T proofs per launch) and verified [REF examples/round.rs:268-350]
  mixed    (config 5) a stream of independent jobs, 50 % prove jobs / 50 % verify jobs (the proofs of the previous step)

`roofline` is measured live with HIP events on the engine's own stream around every kernel launch of the timed region;
`cpu_baseline` times the oracle's single-threaded C++ restatement (arkworks-style algorithms) on a bounded sample of
the same workload on this box's host cores.
"""
import argparse
import glob
import hashlib
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "oracle", "py")):
    if p not in sys.path:
        sys.path.insert(0, p)
# before the HIP runtime starts: enough hardware queues for the engine's two lanes of streams (include/mpshuffle.h: mp_set_pipeline)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
# VALU issue model of gfx950 (measured: tools/microbench/roof.hip -> profiles/r03_roof.json): 256 CUs x 4 SIMDs; a wave64 instruction
# occupies its SIMD's VALU issue port for 2 cycles (v_add/sub/and/or/xor/ashr_i32: 32 lanes per clock) or 4 cycles (every multiply,
# every 64-bit or three-operand integer instruction, shifts, selects: 16 lanes per clock); nothing co-issues.  2.4 GHz = maximum shader
# clock; under this path's kernels the chip settles near 2.0-2.1 GHz at ~1.3 kW (power telemetry in the same file).
SIMDS = 256 * 4
CLK_MAX_GHZ = 2.4
INT_MAD_PEAK_G = 16 * SIMDS * CLK_MAX_GHZ     # v_mad_u64_u32: 16 lanes per clock per SIMD = 39 322 Gmad/s at 2.4 GHz (measured sustained: 35 300)


# ---- distributed helpers (backend-agnostic: RCCL on GPUs, gloo in the CPU tests) -----------------------------
def dist_info():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def bcast_bytes(data, nbytes, src, device):
    """broadcast a byte string from rank `src` to every rank (shared parameters, once per session)"""
    import torch
    import torch.distributed as dist
    if data is not None:
        t = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(device)
    else:
        t = torch.zeros(nbytes, dtype=torch.uint8, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(t, src=src)
    return bytes(t.cpu().numpy().tobytes())


def reduce_max(x, device):
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(x)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def reduce_sum(x, device):
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(x)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_rows(row, device):
    """all-gather one row of float64 per rank (proof counts, verdict sums, seconds) -> list of rows, rank order"""
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(v) for v in row], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(out, t)
        return [[float(v) for v in o.cpu().tolist()] for o in out]
    return [[float(v) for v in t.cpu().tolist()]]


def gather_objects(obj):
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_world_size() > 1:
        out = [None] * dist.get_world_size()
        dist.all_gather_object(out, obj)
        return out
    return [obj]


def rccl_smoke(dist, device):
    """tools/rccl_smoke.py: the collectives this bench needs, run once before anything expensive is built; exits non-zero naming the
    failing call"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("rccl_smoke", os.path.join(ROOT, "tools", "rccl_smoke.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    try:
        return mod.run_checks(dist, device)
    except mod.CollectiveCheckFailed as e:
        print("bench.py: rank %s: collective check failed before any table was built: %s" % (os.environ.get("RANK", "?"), e),
              file=sys.stderr, flush=True)
        os._exit(3)


def pin_to_gpu_numa_node(local):
    """the cpu_baseline leg on the cores next to this rank's GPU: PCI address -> NUMA node -> cpulist (best effort; returns a note)"""
    try:
        import torch
        pr = torch.cuda.get_device_properties(local)
        bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read())
        if node < 0:
            return {"gpu_pci": bdf, "numa_node": None}
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        allowed = os.sched_getaffinity(0) & cpus
        if allowed:
            os.sched_setaffinity(0, allowed)
        return {"gpu_pci": bdf, "numa_node": node, "cpus_pinned": len(allowed) if allowed else 0}
    except Exception as e:           # no sysfs entry, no affinity call: report and carry on unpinned
        return {"numa_node": None, "note": "%s: %s" % (type(e).__name__, e)}


class Telemetry:
    """shader clock and board power of THIS rank's GPU, sampled from its amdgpu hwmon files every 20 ms by a host thread while the timed
    region runs (as tools/microbench/roof.hip does): the first suspects if a multi-GPU line falls short of N x the single-GPU one --
    a node power limit shows as per_rank_power_w / per_rank_sclk_mhz below the N = 1 run's on every rank, a slow rank as one outlier
    (DESIGN.md section 5).  Best effort: None where the sysfs files are missing."""

    def __init__(self, local):
        import glob
        self.power_path = self.sclk_path = None
        self.power, self.sclk = [], []
        self._stop = None
        self._thread = None
        try:
            import torch
            pr = torch.cuda.get_device_properties(local)
            base = "/sys/bus/pci/devices/%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
            for name in ("power1_average", "power1_input"):
                hit = sorted(glob.glob(base + "/hwmon/hwmon*/" + name))
                if hit:
                    self.power_path = hit[0]
                    break
            hit = sorted(glob.glob(base + "/hwmon/hwmon*/freq1_input"))
            self.sclk_path = hit[0] if hit else None
        except Exception:      # noqa: BLE001 -- telemetry never fails the run
            pass

    @staticmethod
    def _read(path):
        try:
            return float(open(path).read().strip())
        except Exception:      # noqa: BLE001
            return None

    def start(self):
        import threading
        if not (self.power_path or self.sclk_path):
            return
        self._stop = threading.Event()

        def loop():
            while not self._stop.is_set():
                p = self._read(self.power_path) if self.power_path else None
                f = self._read(self.sclk_path) if self.sclk_path else None
                if p is not None:
                    self.power.append(p * 1e-6)      # microwatts
                if f is not None:
                    self.sclk.append(f * 1e-6)       # hertz
                self._stop.wait(0.02)
        self._thread = threading.Thread(target=loop, daemon=True)
        self._thread.start()

    def stop(self):
        if self._thread:
            self._stop.set()
            self._thread.join()
        mean = lambda v: (sum(v) / len(v)) if v else -1.0
        return mean(self.sclk), mean(self.power), float(len(self.power) or len(self.sclk))


def scaled_batch(batch, world, scaling, seed_block):
    """proofs per rank per step and the seeding block: weak = `batch` each; strong = `batch` in total, a contiguous 1/N per rank, seeded
    per rank-sized block of the global batch so that the bytes are those of the unsharded run of the same proofs"""
    if scaling == "weak":
        return batch, seed_block
    if batch % world:
        raise SystemExit("--scaling strong: --batch (%d) must be a multiple of the number of ranks (%d)" % (batch, world))
    per = batch // world
    if seed_block is None:
        seed_block = per
    if per % seed_block:
        raise SystemExit("--scaling strong: --seed-block (%d) must divide the per-rank batch (%d)" % (seed_block, per))
    return per, seed_block


def shard_range(total, rank, world):
    """static contiguous block partition of proof indices"""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n_ranks, argv):
    """plain-shell start with --gpus N > 1: run N ranks of this script under torch.distributed.run (one per GPU)"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


# ---- reproducible measurement metadata ------------------------------------------------------------------------------
def engine_source_hash():
    """sha256 over the engine's sources: ties a PMC summary under profiles/ to the build it was taken from"""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "mental-poker_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hpp", ".hip")):
            h.update(f.encode())
            h.update(open(os.path.join(csrc, f), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "mpshuffle.h"), "rb").read())
    return h.hexdigest()[:16]


def find_pmc_summary(src_hash, curve, m, n, batch, workload):
    """the PMC summary (tools/pmc_summary.py) taken from THIS build at THIS configuration, or None"""
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_summary*.json")), reverse=True):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        if (d.get("engine_src") == src_hash and d.get("batch") == batch and d.get("curve") == curve and
                d.get("m") == m and d.get("n") == n and d.get("workload", "pairs") == workload):
            return os.path.relpath(path, ROOT), d
    return None, None


def a1_of(mc):
    return mc.get("a1", 0)


def load_mad_counts(curve):
    """multiply-add instructions (v_mad_u64_u32 + v_mad_i64_i32) per field product / square / fused product pair of this
    curve's base field, counted in the gfx950 assembly of the build (tools/gen_mad_counts.py -> mad_counts.json)"""
    try:
        d = json.load(open(os.path.join(ROOT, "mental-poker_amd", "mad_counts.json")))
    except (OSError, ValueError):
        return None
    c = d.get("curves", {}).get(curve)
    if not c:
        return None
    M, S, MS = c["mul"], c["sqr"], c["mulsub"]
    inv = c.get("inv_divsteps_mads", c["inv_sqr"] * c["sqr"] + c["inv_mul"] * c["mul"])
    a1 = 1 if c["a_is_one"] else 0
    return {
        "issue": c.get("valu_issue"),
        "field": {"mul": M, "sqr": S, "mulsub": MS, "inv": inv},
        "madd": 6 * M + 2 * S + MS,                      # XYZZ += affine   (curve.hpp xyzz_madd_ip, main path)
        "dbl": 4 * M + (3 + a1) * S + MS,                # XYZZ doubling    (xyzz_dbl_ip)
        "aff": 5 * M + (1 + 8.0 / 15.0) * S + (4.0 / 15.0) * inv / 64.0,   # batched-affine table entry (k_table): 2 prefix-product steps, lam,
                                                         # lam^2, y3 (+ x^2 for the 8 doublings of 15); 4 inversions per 64 bases
        "jac": 11 * M + 5 * S,                           # Jacobian + Jacobian (k_combine)
        "a1": a1,
        "norm": 6 * M + S + inv / 64.0,                  # Jacobian -> affine with a 64-point batched inversion
    }


# ---- the benchmark -----------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=None,
                    help="proofs per GPU per step (default 262144 = ~0.45 MB of HBM each: 142 GB in use; chain32: card tables per GPU)")
    ap.add_argument("--workload", default="pairs", choices=["pairs", "chain32", "mixed"])
    ap.add_argument("--m", type=int, default=2)
    ap.add_argument("--n", type=int, default=26)
    ap.add_argument("--curve", default=None, help="stark (default; mixed: secp256k1), bn254, secp256k1, bls12_377")
    ap.add_argument("--streams", type=int, default=1, help="independent engine contexts (HIP streams) per GPU; the batch is split evenly")
    ap.add_argument("--fb-bits", type=int, default=None, help="fixed-base window width (8, 16, 20 or 21 bits; 20 = 27 GB of tables at n=26, 21 = 48 GB and 12 instead of 13 windows on the STARK curve; default: 21 on the STARK curve, 20 elsewhere)")
    ap.add_argument("--cpu-iters", type=int, default=160, help="prove+verify pairs timed for cpu_baseline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the two extra one-step measurements (keyed batch, per-equation verification) reported in config")
    ap.add_argument("--latency-batch", type=int, default=None,
                    help="batches up to this size use the latency plan (engine default 2560 x 52 / N): mp_set_latency_batch")
    ap.add_argument("--keyed", type=int, default=0, metavar="K",
                    help="keyed batches: K distinct aggregate keys (card tables) spread over the batch, one key per proof "
                         "(mp_*_batch_keys_dev); 0 = every proof under the table's own key")
    ap.add_argument("--bucket-min", type=int, default=None,
                    help="variable-base MSMs of at least this many terms run on the bucket-method kernel (engine default 2048; 0 = never)")
    ap.add_argument("--bucket-bits", type=int, default=None, help="window width of the bucket method (8 .. 13; default: by the size of the MSM)")
    ap.add_argument("--group-points", type=int, default=None,
                    help="points per group equation of the verifier's screen (default 243712 = 1 024 proofs of a 52-card deck on the split bucket pipeline; "
                         "30464 = the 128 proofs of rounds 4-5 on one wave per window; 0 = per-proof screen)")
    ap.add_argument("--transcript-lanes", type=int, default=None, choices=[0, 1, 4],
                    help="lanes per Fiat-Shamir transcript hash (mp_set_transcript_lanes; default: by batch size)")
    ap.add_argument("--group-lanes", type=int, default=None, choices=[0, 1, 4],
                    help="lanes per group operation of the MSM chains (mp_set_group_lanes; default: by batch size)")
    ap.add_argument("--work-split", type=int, default=None, choices=[-1, 0, 1, 2, 3, 4, 5],
                    help="force one of the table's work splits (mp_set_work_split; default: by batch size)")
    ap.add_argument("--per-equation", action="store_true", help="verify equation by equation (mp_set_merged_verify off)")
    ap.add_argument("--players", type=int, default=32, help="chain32: shuffles per table")
    ap.add_argument("--no-keyset", action="store_true",
                    help="chain32 / --keyed: pass the aggregate keys with every call (mp_*_batch_keys_dev: the key's window tables are rebuilt "
                         "per proof) instead of preparing them once as a key set (mp_keyset_create, built outside the timed region like "
                         "the fixed-base tables of the shared parameters)")
    ap.add_argument("--per-link-verify", action="store_true",
                    help="chain32: verify every link on its own (mp_verify_shuffle_batch_keys_dev) instead of one chain equation per table "
                         "(mp_verify_shuffle_chain_dev)")
    ap.add_argument("--no-subgroup-check", action="store_true",
                    help="curves with a cofactor (bls12_377): skip the per-point subgroup test of the wire points (mp_set_subgroup_check off) -- the "
                         "reference's shuffle_and_remask / verify_shuffle take typed points that were validated when they were deserialised "
                         "[REF examples/parameter_selection.rs:78-91], so this is its like-for-like; the default keeps the test")
    ap.add_argument("--validated-once", action="store_true",
                    help="pairs workload: every deck is validated exactly ONCE inside the timed step (mp_deck_validate_dev on the input decks and on "
                         "the shuffled decks, as the party that receives them would) and the prove / verify calls are told so (mp_set_validated): the "
                         "proofs' points are still tested in the verify call -- every point of a pair is tested once instead of the decks in every "
                         "call that touches them (curves with a cofactor: bls12_377)")
    ap.add_argument("--chain-max-links", type=int, default=None,
                    help="chain32: links per chain equation (mp_set_chain_max_links; a chain of --players links is verified as consecutive "
                         "sub-chains; default: the whole chain)")
    ap.add_argument("--chain-slice", type=int, default=None,
                    help="chain32: tables per pass of chain verification (mp_set_chain_slice; default: the library's rule -- one pass if the "
                         "workspace fits the free memory)")
    ap.add_argument("--chain-group", type=int, default=None,
                    help="chain32: tables per chain equation (mp_set_chain_group; default: by size, 1 = every table on its own)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = --batch proofs per GPU per step (default); strong = --batch proofs per step in total, 1/N per GPU")
    ap.add_argument("--pipeline", type=int, default=0,
                    help="pairs: mp_set_pipeline depth for the timed region (verify calls on the engine's second lane; 0 = off, the default; "
                         "the extra `batch_curve` always reports both)")
    ap.add_argument("--digest", action="store_true",
                    help="test hook: sha256 of every rank's outputs of the last step in config.digests (inputs are seeded per "
                         "block of --seed-block proofs, so a sharded run and an unsharded run of the same blocks must agree)")
    ap.add_argument("--seed-block", type=int, default=None)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))

    # (the host driver only supports dmabuf IPC: without this RCCL fails with hipIpcGetMemHandle; exported on the GPU boxes, set here too)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    rank, world, local = dist_info()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
    # test hooks (used to dry-run the N > 1 path on a single-GPU box): MP_BENCH_FORCE_DEVICE puts every rank on one GPU,
    # MP_BENCH_BACKEND=gloo runs the (tiny, once-per-session) collectives on CPU tensors instead of RCCL
    local = int(os.environ.get("MP_BENCH_FORCE_DEVICE", local))
    backend = os.environ.get("MP_BENCH_BACKEND", "nccl")
    if local >= torch.cuda.device_count():
        raise SystemExit("rank %d: no GPU %d on this node (%d visible)" % (rank, local, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    gpu = torch.device("cuda", local)
    dev = gpu if backend == "nccl" else torch.device("cpu")      # device of the collective payloads
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=gpu)
        else:
            dist.init_process_group(backend)
    if world != args.gpus and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    smoke_log = rccl_smoke(dist, dev) if world > 1 else None      # fails fast, before any table is built

    mp = importlib.import_module("mental-poker_amd")
    workload = args.workload
    curve = args.curve or ("secp256k1" if workload == "mixed" else "stark")
    m, n = args.m, args.n
    N = m * n
    B = args.batch if args.batch is not None else (65536 if workload == "chain32" else 262144)
    # (--scaling strong: the batch is the WHOLE job's, every rank takes its contiguous 1/N of the proof indices)
    B, args.seed_block = scaled_batch(B, world, args.scaling, args.seed_block)
    if args.fb_bits is None:
        # widest fixed-base windows whose tables ((n + 5) bases x windows x (2^bits - 1) entries) stay below 30 % of this GPU's HBM:
        # 21 bits (48 GB) at n = 26 on the STARK curve, 20 bits elsewhere, 16 bits for the 1024-card shapes (n >= 64)
        hbm_total = torch.cuda.mem_get_info(local)[1]
        sbits = {"stark": 252, "bn254": 254, "secp256k1": 256, "bls12_377": 253}[curve]
        pbytes = 96 if curve == "bls12_377" else 64
        args.fb_bits = 8
        for bits in ((21, 20, 16) if curve == "stark" else (20, 16)):
            if (n + 5) * ((sbits + bits - 1) // bits) * ((1 << bits) - 1) * pbytes <= 0.30 * hbm_total:
                args.fb_bits = bits
                break
    eng = mp.Engine(curve, device=local)
    PB = eng.point_bytes
    CB = 2 * PB

    # ---- shared parameters: rank 0 runs `setup`, everyone receives them over RCCL (once)
    psz = PB * (n + 3)
    blob = None
    if rank == 0:
        params = eng.setup(m, n, bytes([1] * 32))
        pk = eng.setup(m, 2, bytes([2] * 32))[:PB]                  # aggregate key: a random group element
        base_deck = eng.setup(m, 2 * N - 3, bytes([3] * 32))        # 2N random points = N random ciphertexts
        blob = params + pk + base_deck
    blob = bcast_bytes(blob, psz + PB + CB * N, 0, dev)
    params, pk, base_deck = blob[:psz], blob[psz:psz + PB], blob[psz + PB:]
    # ---- S independent contexts (one HIP stream each); each owns 1/S of the batch
    S = max(1, args.streams) if workload == "pairs" else 1
    Bs = B // S
    assert Bs * S == B, "--batch must be a multiple of --streams"
    engines = [eng] + [mp.Engine(curve, device=local) for _ in range(S - 1)]
    t_tab = time.perf_counter()
    tables = [e.table(m, n, params, pk, fb_bits=args.fb_bits) for e in engines]
    for e in engines:
        e.sync()
    table_build_s = time.perf_counter() - t_tab      # fixed-base window tables of the shared parameters: built once per session, outside the timed region
    for t in tables:
        if args.latency_batch is not None:
            t.set_latency_batch(args.latency_batch)
        if args.per_equation:
            t.set_merged_verify(False)
        if args.bucket_min is not None:
            t.set_bucket_min(args.bucket_min)
        if args.bucket_bits is not None:
            t.set_bucket_bits(args.bucket_bits)
        if args.group_points is not None:
            t.set_group_verify(args.group_points, 6144)
        if args.transcript_lanes is not None:
            t.set_transcript_lanes(args.transcript_lanes)
        if args.work_split is not None:
            t.set_work_split(args.work_split)
        if args.group_lanes is not None:
            t.set_group_lanes(args.group_lanes)
        if args.no_subgroup_check:
            t.set_subgroup_check(False)
        if args.validated_once:
            t.set_validated(t.VALIDATED_DECKS | t.VALIDATED_SHUFFLED)
    table = tables[0]
    proof_bytes = table.proof_bytes

    # ---- synthetic inputs, resident in HBM; seeded per block of `seed_block` proofs of the GLOBAL batch (rank r owns the
    # proofs [r B, (r+1) B) of the world x B proofs of one step), so the data do not depend on how the batch is sharded
    seed_block = args.seed_block or B
    assert B % seed_block == 0
    gen = torch.Generator(device=gpu)

    def per_block(fn, salt):
        outs = []
        for k in range(B // seed_block):
            gen.manual_seed(1234 + 7919 * salt + (rank * B) // seed_block + k)
            outs.append(fn(seed_block))
        return torch.cat(outs).contiguous() if len(outs) > 1 else outs[0].contiguous()

    def rand_bytes(cnt, *shape):
        return torch.randint(0, 256, (cnt,) + shape, dtype=torch.uint8, device=gpu, generator=gen)

    def rand_factors(salt):
        def f(cnt):
            x = rand_bytes(cnt, N, 32)
            x[:, :, 31] &= 0x07            # < 2^251 < group order
            return x
        return per_block(f, salt)

    def rand_perms(salt):
        return per_block(lambda cnt: torch.argsort(torch.rand(cnt, N, device=gpu, generator=gen), dim=1).to(torch.int32), salt)

    def rand_seeds(salt):
        return per_block(lambda cnt: rand_bytes(cnt, 32), salt)

    def sync_all():
        for e in engines:
            e.sync()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        sync_all()

    # keyed batches: K aggregate keys (random group elements from the engine's own setup), key b % K for proof b
    def make_keys(K, count):
        kpts = eng.setup(m, max(K, 2), bytes([4] * 32))
        kt = torch.frombuffer(bytearray(kpts[:PB * K]), dtype=torch.uint8).to(gpu).view(K, PB)
        return kt[torch.arange(count, device=gpu) % K].contiguous()

    def make_keyset(tbl, K, count):
        """the same K keys as make_keys as a key set of `tbl`, and the per-proof key indices (b % K)"""
        kpts = eng.setup(m, max(K, 2), bytes([4] * 32))
        t0 = time.perf_counter()
        ks = tbl.keyset(kpts[:PB * K])
        eng.sync()
        return ks, (torch.arange(count, device=gpu) % K).to(torch.int32).contiguous(), time.perf_counter() - t0

    base = torch.frombuffer(bytearray(base_deck), dtype=torch.uint8).to(gpu)

    def prime_decks(count, keys=None):
        """`count` different random decks = re-encryptions of the base deck (untimed input generation)"""
        d0 = base.repeat(count, 1).contiguous()
        out = torch.empty(count, N * CB, dtype=torch.uint8, device=gpu)
        pr = torch.empty(count, proof_bytes, dtype=torch.uint8, device=gpu)
        st = torch.empty(count, dtype=torch.int32, device=gpu)
        f0, p0, s0 = rand_factors(101), rand_perms(102), rand_seeds(103)
        torch.cuda.synchronize()
        table.shuffle_and_remask_batch_dev(count, d0.data_ptr(), f0[:count].data_ptr(), p0[:count].data_ptr(),
                                           s0[:count].data_ptr(), out.data_ptr(), pr.data_ptr(), st.data_ptr())
        eng.sync()
        assert int(st.abs().sum().item()) == 0, "priming pass failed"
        return out

    digests = None
    extras = {}
    # =============================================================================================== workload: pairs
    if workload == "pairs":
        for t in tables:
            t.reserve(Bs)
        keys = make_keys(args.keyed, B) if args.keyed > 0 else None
        for e in engines:
            e.profile_enable(True)
        decks = prime_decks(B)
        priming_launches = {}       # per-kernel launches of the untimed priming prove (tools/pmc_summary.py skips them)
        for e in engines:
            for k, (cnt, _) in e.profile_report().items():
                priming_launches[k] = priming_launches.get(k, 0) + cnt
            e.profile_enable(False)
        # (--pipeline D: a verify call's inputs stay untouched until D further verify calls have returned -> D + 1 sets of prover outputs)
        out_sets = [(torch.empty(B, N * CB, dtype=torch.uint8, device=gpu), torch.empty(B, proof_bytes, dtype=torch.uint8, device=gpu),
                     torch.empty(B, dtype=torch.int32, device=gpu)) for _ in range(max(args.pipeline, 0) + 1)]
        out_decks, out_proofs, st_v = out_sets[0]
        st_p = torch.empty(B, dtype=torch.int32, device=gpu)
        st_val = torch.zeros(B, dtype=torch.int32, device=gpu)       # --validated-once: the verdicts of mp_deck_validate_dev
        st_val2 = torch.zeros(B, dtype=torch.int32, device=gpu)
        factors, perms, seeds = rand_factors(1), rand_perms(2), rand_seeds(3)
        torch.cuda.synchronize()
        if args.pipeline > 0:
            assert S == 1 and args.keyed == 0, "--pipeline: one context, the table's own key"
            table.set_pipeline(args.pipeline)
        rot = {"i": 0}

        def sl(t, i):
            return t[i * Bs:(i + 1) * Bs].data_ptr()

        kset = None
        if keys is not None and not args.no_keyset and S == 1:
            kset = make_keyset(table, args.keyed, B)
            extras["keyset_build_s"] = kset[2]

        def step(kk=keys, kset=kset):
            for i, t in enumerate(tables):
                if kset is not None:
                    t.shuffle_and_remask_batch_keyset_dev(kset[0], Bs, kset[1].data_ptr(), sl(decks, i), sl(factors, i), sl(perms, i), sl(seeds, i),
                                                          sl(out_decks, i), sl(out_proofs, i), sl(st_p, i))
                    t.verify_shuffle_batch_keyset_dev(kset[0], Bs, kset[1].data_ptr(), sl(decks, i), sl(out_decks, i), sl(out_proofs, i), sl(st_v, i))
                    continue
                if kk is not None:
                    t.shuffle_and_remask_batch_keys_dev(Bs, sl(kk, i), sl(decks, i), sl(factors, i), sl(perms, i), sl(seeds, i),
                                                        sl(out_decks, i), sl(out_proofs, i), sl(st_p, i))
                    t.verify_shuffle_batch_keys_dev(Bs, sl(kk, i), sl(decks, i), sl(out_decks, i), sl(out_proofs, i), sl(st_v, i))
                    continue
                od, op_, sv = out_sets[rot["i"]]
                rot["i"] = (rot["i"] + 1) % len(out_sets)
                if args.validated_once:       # the input decks, once (the party that received them)
                    t.deck_validate_dev(Bs, sl(decks, i), sl(st_val, i))
                t.shuffle_and_remask_batch_dev(Bs, sl(decks, i), sl(factors, i), sl(perms, i), sl(seeds, i), sl(od, i), sl(op_, i), sl(st_p, i))
                if args.validated_once:       # the shuffled decks, once (the verifier that receives them with the proof)
                    t.deck_validate_dev(Bs, sl(od, i), sl(st_val2, i))
                t.verify_shuffle_batch_dev(Bs, sl(decks, i), sl(od, i), sl(op_, i), sl(sv, i))

        proofs_per_step = B
        units = "prove+verify pairs"

        def check():
            return (int((st_p != 0).sum().item()) + sum(int((o[2] != 0).sum().item()) for o in out_sets) +
                    (int((st_val != 0).sum().item()) + int((st_val2 != 0).sum().item()) if args.validated_once else 0))

        def parity_inputs():
            b = B // 2
            pk_b = pk if keys is None else bytes(keys[b].cpu().numpy().tobytes())
            return (pk_b, decks[b], factors[b], perms[b], seeds[b], out_decks[b], out_proofs[b])

        def digest():
            h = []
            for k in range(B // seed_block):
                s_ = slice(k * seed_block, (k + 1) * seed_block)
                hh = hashlib.sha256()
                hh.update(out_decks[s_].cpu().numpy().tobytes())
                hh.update(out_proofs[s_].cpu().numpy().tobytes())
                h.append(hh.hexdigest())
            return h
    # =============================================================================================== workload: chain32
    elif workload == "chain32":
        # T tables per GPU, `players` dependent shuffles each; table t has its own aggregate key (keyed batches); link j of
        # every table is proved in ONE launch of T proofs (the data-parallel axis runs across tables), all decks stay in HBM.
        T, L = B, args.players
        keyless = eng.table(m, n, params, None, fb_bits=args.fb_bits)
        if args.latency_batch is not None:
            keyless.set_latency_batch(args.latency_batch)
        keys = make_keys(T, T)                 # one key per table
        deck0 = prime_decks(T)
        table.close()
        chain = torch.empty(L + 1, T, N * CB, dtype=torch.uint8, device=gpu)
        chain[0] = deck0
        del deck0
        proofs = torch.empty(L, T, proof_bytes, dtype=torch.uint8, device=gpu)
        st_p = torch.empty(L, T, dtype=torch.int32, device=gpu)
        st_v = torch.empty(L, T, dtype=torch.int32, device=gpu)
        # every link has its own witness AND its own prover seed: a prover seed must never be reused with a different witness
        # (include/mpshuffle.h, mp_shuffle_and_remask: two proofs from one seed reveal the permutation and the masking factors)
        fac = [rand_factors(1000 + j) for j in range(L)]
        prm = [rand_perms(2000 + j) for j in range(L)]
        sds = [rand_seeds(3000 + j) for j in range(L)]
        # verification: one chain equation per table over all L links (default), or link by link in batches of G links
        chain_verify = not args.per_link_verify
        G = L if chain_verify else max(1, min(L, 262144 // T))
        while L % G:
            G -= 1
        kk = keys.repeat(G, 1).contiguous()
        keyless.reserve(T if chain_verify else max(T, G * T))
        chain_links = args.chain_max_links if args.chain_max_links is not None else L
        if chain_verify and chain_links < L:
            keyless.set_chain_max_links(chain_links)
        if args.chain_slice is not None:
            keyless.set_chain_slice(args.chain_slice)
        if args.chain_group is not None:
            keyless.set_chain_group(args.chain_group)
        extras["chain_links_per_equation"] = min(L, chain_links) if chain_verify else None
        kset = None
        if not args.no_keyset:
            kset = make_keyset(keyless, T, T)       # one key per table, prepared once (a table keeps its key across hands)
            extras["keyset_build_s"] = kset[2]
        priming_launches = {}
        torch.cuda.synchronize()

        def step():
            for j in range(L):
                if kset is not None:
                    keyless.shuffle_and_remask_batch_keyset_dev(kset[0], T, kset[1].data_ptr(), chain[j].data_ptr(), fac[j].data_ptr(),
                                                                prm[j].data_ptr(), sds[j].data_ptr(), chain[j + 1].data_ptr(),
                                                                proofs[j].data_ptr(), st_p[j].data_ptr())
                    continue
                keyless.shuffle_and_remask_batch_keys_dev(T, keys.data_ptr(), chain[j].data_ptr(), fac[j].data_ptr(),
                                                          prm[j].data_ptr(), sds[j].data_ptr(), chain[j + 1].data_ptr(),
                                                          proofs[j].data_ptr(), st_p[j].data_ptr())
            if chain_verify:
                keyless.verify_shuffle_chain_dev(T, L, kk.data_ptr(), chain.data_ptr(), proofs.data_ptr(), st_v.data_ptr())
                return
            for j in range(0, L, G):
                keyless.verify_shuffle_batch_keys_dev(G * T, kk.data_ptr(), chain[j].data_ptr(), chain[j + 1].data_ptr(),
                                                      proofs[j].data_ptr(), st_v[j].data_ptr())

        proofs_per_step = T * L
        units = ("prove+verify pairs (%d tables x %d dependent shuffles, %d proofs per prove launch; keys: %s; verification: %s)"
                 % (T, L, T, "one key set of %d keys prepared before the timed region (mp_keyset_create)" % T if kset is not None
                    else "passed with every call", "chain equations over all links, the chains of several tables per equation (mp_verify_shuffle_chain_dev; config.chain_tables_per_equation)" if chain_verify
                    else "%d proofs per launch, link by link" % (G * T)))

        def check():
            return int((st_p != 0).sum().item()) + int((st_v != 0).sum().item())

        def parity_inputs():
            t_, j = T // 2, L - 1
            return (bytes(keys[t_].cpu().numpy().tobytes()), chain[j][t_], fac[j][t_], prm[j][t_], sds[j][t_],
                    chain[j + 1][t_], proofs[j][t_])

        def digest():
            hh = hashlib.sha256()
            hh.update(chain[L].cpu().numpy().tobytes())
            return [hh.hexdigest()]
        table = keyless
        tables = [keyless]
    # =============================================================================================== workload: mixed
    else:
        # a stream of B independent jobs per step: B/2 prove jobs (fresh decks) and B/2 verify jobs (the proofs made by the
        # previous step's prove jobs) -- BASELINE config 5
        H = B // 2
        table.reserve(H)
        decks = prime_decks(H)
        out_decks = [torch.empty(H, N * CB, dtype=torch.uint8, device=gpu) for _ in range(2)]
        out_proofs = [torch.empty(H, proof_bytes, dtype=torch.uint8, device=gpu) for _ in range(2)]
        st_p = torch.empty(H, dtype=torch.int32, device=gpu)
        st_v = torch.zeros(H, dtype=torch.int32, device=gpu)
        factors, perms, seeds = rand_factors(1)[:H], rand_perms(2)[:H], rand_seeds(3)[:H]
        priming_launches = {}
        state = {"cur": 0, "have_prev": False}
        torch.cuda.synchronize()

        def step():
            c = state["cur"]
            table.shuffle_and_remask_batch_dev(H, decks.data_ptr(), factors.data_ptr(), perms.data_ptr(), seeds.data_ptr(),
                                               out_decks[c].data_ptr(), out_proofs[c].data_ptr(), st_p.data_ptr())
            p = c if not state["have_prev"] else 1 - c       # very first step: verify what was just proved
            table.verify_shuffle_batch_dev(H, decks.data_ptr(), out_decks[p].data_ptr(), out_proofs[p].data_ptr(), st_v.data_ptr())
            state["cur"], state["have_prev"] = 1 - c, True

        proofs_per_step = H
        units = "pair equivalents: %d prove jobs + %d verify jobs per step" % (H, H)

        def check():
            return int((st_p != 0).sum().item()) + int((st_v != 0).sum().item())

        def parity_inputs():
            b, c = H // 2, 1 - state["cur"]
            return (pk, decks[b], factors[b], perms[b], seeds[b], out_decks[c][b], out_proofs[c][b])

        def digest():
            hh = hashlib.sha256()
            hh.update(out_proofs[1 - state["cur"]].cpu().numpy().tobytes())
            return [hh.hexdigest()]

    # ---- timed region: W warm-up steps, barrier + synchronize, K steps, barrier + synchronize
    for _ in range(args.warmup):
        step()
    barrier()
    for e in engines:
        e.profile_enable(True)
    tele = Telemetry(local)
    tele.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    my_enqueue = time.perf_counter() - t0      # every call of the K steps has returned (a verify call returns with its screening verdict)
    barrier()
    my_elapsed = time.perf_counter() - t0
    my_sclk, my_power, my_samples = tele.stop()
    prof = {}
    for e in engines:
        for k, (cnt, ms) in e.profile_report().items():
            c0, m0 = prof.get(k, (0, 0.0))
            prof[k] = (c0 + cnt, m0 + ms)
        e.profile_enable(False)
    elapsed = reduce_max(my_elapsed, dev)
    hbm_used_gb = round((torch.cuda.mem_get_info(local)[1] - torch.cuda.mem_get_info(local)[0]) / 1e9, 1)

    # ---- correctness of what was timed (outside the timed region); verdicts gathered from every rank
    bad = check()
    rows = gather_rows([proofs_per_step * args.steps, bad, my_elapsed, my_sclk, my_power, my_samples, table_build_s, my_enqueue], dev)
    total_proofs = sum(r[0] for r in rows)
    total_bad = int(sum(r[1] for r in rows))
    assert total_bad == 0, "%d proofs failed (per rank: %s)" % (total_bad, [int(r[1]) for r in rows])
    if args.digest:
        digests = gather_objects(digest())
    parity = None
    if rank == 0:
        import coracle as co
        co.build()
        pk_b, d_in, f_in, p_in, s_in, d_out, pr_out = parity_inputs()
        tb = lambda t: bytes(t.cpu().numpy().tobytes())
        exp_deck, exp_proof = co.shuffle_and_remask(curve, m, n, params, pk_b, tb(d_in), tb(f_in), [int(v) for v in p_in.cpu().tolist()], tb(s_in))
        parity = tb(d_out) == exp_deck and tb(pr_out) == exp_proof
        assert parity, "timed output differs from the oracle"
    value = total_proofs / elapsed

    # ---- the conditions of the headline, measured beside it (pairs workload, N = 1): one step each, same batch
    if workload == "pairs" and world == 1 and not args.no_extras and args.keyed == 0 and not args.per_equation and S == 1:
        def timed_step(fn):
            fn()
            barrier()
            t1 = time.perf_counter()
            fn()
            barrier()
            return B / (time.perf_counter() - t1)
        # proof bytes of the timed region at the rows the extras below compare against (the keyed steps overwrite the output buffers)
        ref_rows = {i: out_proofs[i].clone() for i in {512, 2048, 8192, 16384, B // 2} if i < B}
        table.set_merged_verify(False)
        extras["per_equation_value"] = timed_step(step)          # every equation its own MSM, as the reference evaluates them
        table.set_merged_verify(True)
        if table.group_size(Bs):
            table.set_group_verify(0, 0)
            extras["per_proof_screen_value"] = timed_step(step)  # one merged equation per PROOF (Straus), the headline of rounds 1-3
            table.set_group_verify(args.group_points if args.group_points is not None else 243712, 6144)      # (back to the run's own setting: 243 712 = the engine's default)
        # ---- what a rejected proof costs.  The same step with tampered proofs in the batch: the prover runs as in
        # the timed region, one byte of the last response scalar of the chosen proofs is flipped in HBM, the verifier must reject exactly
        # those (by the name of their first failing check) and accept the rest.  one_bad: 1 proof of the batch; pct1_bad: 1 % of it,
        # evenly spread.  A failing group's members are looked at again, nobody else (mp_set_group_refine).
        def rejection(Bc, nbad, warm=1):
            # (WHICH proofs: a seeded random choice -- a regular stride lands in a quarter of the groups, lane of (member j, group t) = j T + t)
            gsel = torch.Generator(device="cpu")
            gsel.manual_seed(1000003 * Bc + nbad)
            idx = torch.randperm(Bc, generator=gsel)[:nbad].sort().values.to(gpu)
            od, op_, sv = out_sets[0]

            def one():
                table.shuffle_and_remask_batch_dev(Bc, decks.data_ptr(), factors.data_ptr(), perms.data_ptr(), seeds.data_ptr(),
                                                   od.data_ptr(), op_.data_ptr(), st_p.data_ptr())
                eng.sync()
                op_[idx, proof_bytes - 31] ^= 2
                torch.cuda.synchronize()
                table.verify_shuffle_batch_dev(Bc, decks.data_ptr(), od.data_ptr(), op_.data_ptr(), sv.data_ptr())
                eng.sync()
            first = None
            for _ in range(warm):
                t1 = time.perf_counter()
                one()
                first = first or round(Bc / (time.perf_counter() - t1), 1)
            looked = table.reverified_count()
            reps = 2 if Bc >= 65536 else 8
            t1 = time.perf_counter()
            for _ in range(reps):
                one()
            dt = time.perf_counter() - t1
            looked = (table.reverified_count() - looked) // reps
            want = torch.zeros(Bc, dtype=torch.bool, device=gpu)
            want[idx] = True
            assert torch.equal(sv[:Bc] != 0, want), "rejection: the verifier did not reject exactly the tampered proofs"
            assert nbad == 0 or (int((sv[:Bc][idx] != sv[idx[0]]).sum().item()) == 0 and int(sv[idx[0]].item()) > 0)
            return round(Bc * reps / dt, 1), int(looked), first
        gl_honest = table.group_size(B)
        if not args.pipeline:
            for tag, Bc in (("", B), ("_16384", 16384)):
                if Bc > B or (tag and Bc == B):
                    continue
                # (untimed: the buffers of the finer passes exist before anything below is measured -- a table allocates them once in its
                # life, and `pct1_bad_first_value` is about what the groups do, not about hipMalloc; then honest traffic restores the groups)
                rejection(Bc, max(1, Bc // 100), warm=1)
                for _ in range(6):
                    step()
                    eng.sync()
                assert table.group_size(B) == gl_honest
                if tag:
                    extras["none_bad%s_value" % tag] = rejection(Bc, 0)[0]      # the same calls, nobody tampered with: what the two below compare to
                extras["one_bad%s_value" % tag], extras["one_bad%s_reverified" % tag], _ = rejection(Bc, 1, warm=2)
                # 1 % of the traffic tampered with, sustained: the table's groups adapt to what its screens see (mp_set_group_adapt) --
                # `pct1_bad_value` is the rate it settles at (the 5th call on), `pct1_bad_first_value` the first such call after honest traffic
                (extras["pct1_bad%s_value" % tag], extras["pct1_bad%s_reverified" % tag],
                 extras["pct1_bad%s_first_value" % tag]) = rejection(Bc, max(1, Bc // 100), warm=4)
                extras["pct1_bad%s_group_size" % tag] = table.group_size(Bc)
                for _ in range(5):                                 # honest traffic again: the groups grow back, one step per call
                    step()
                    eng.sync()
                assert table.group_size(B) == gl_honest
            step()                                                # (the output buffers hold honest proofs again)
            eng.sync()
        free_b, _ = torch.cuda.mem_get_info()
        if free_b > 40e9:
            k1000 = make_keys(1000, B)
            extras["keyed_value"] = timed_step(lambda: step(k1000, None))   # one aggregate key per proof (1000 distinct), as the reference passes it per call
            extras["keyed_distinct_keys"] = 1000
            ks1000 = make_keyset(table, 1000, B)                      # the same keys prepared once as a key set (mp_keyset_create)
            extras["keyed_keyset_value"] = timed_step(lambda: step(None, ks1000))
            extras["keyset_build_s"] = ks1000[2]
            ks1000[0].close()
        else:
            extras["keyed_value"] = None
        assert check() == 0
        # ---- the throughput at the batch sizes a card server actually has in flight.  Same table, the first Bc proofs
        # of the same inputs; the engine picks its work split by batch size (include/mpshuffle.h: mp_set_latency_batch).  `serial`: the calls
        # as in the timed region (prove, then verify, one lane); `pipelined`: mp_set_pipeline(1) -- the verify call of batch k runs on the
        # engine's second lane beside the prove call of batch k + 1 (two sets of prover outputs, verdicts examined one call later)
        def curve_point(Bc, depth, seconds=0.5):
            sets = [(torch.empty(Bc, N * CB, dtype=torch.uint8, device=gpu), torch.empty(Bc, proof_bytes, dtype=torch.uint8, device=gpu),
                     torch.zeros(Bc, dtype=torch.int32, device=gpu)) for _ in range(depth + 1)]
            stp = torch.zeros(Bc, dtype=torch.int32, device=gpu)
            torch.cuda.synchronize()
            table.set_pipeline(depth)
            state = {"i": 0}

            def one():
                od, op_, sv = sets[state["i"]]
                state["i"] = (state["i"] + 1) % len(sets)
                table.shuffle_and_remask_batch_dev(Bc, decks.data_ptr(), factors.data_ptr(), perms.data_ptr(), seeds.data_ptr(),
                                                   od.data_ptr(), op_.data_ptr(), stp.data_ptr())
                table.verify_shuffle_batch_dev(Bc, decks.data_ptr(), od.data_ptr(), op_.data_ptr(), sv.data_ptr())
            one()
            eng.sync()
            t1 = time.perf_counter()
            one()
            eng.sync()
            k = max(2, min(400, int(seconds / max(time.perf_counter() - t1, 1e-4))))
            k += (-k) % len(sets)
            t1 = time.perf_counter()
            for _ in range(k):
                one()
            eng.sync()
            dt = time.perf_counter() - t1
            table.set_pipeline(0)
            assert int((stp != 0).sum().item()) + sum(int((o[2] != 0).sum().item()) for o in sets) == 0, "batch_curve: a proof failed"
            assert torch.equal(sets[0][1][Bc // 2], ref_rows[Bc // 2]), "batch_curve: proof bytes depend on the batch size"
            return round(Bc * k / dt, 1)
        if B >= 32768 and not args.pipeline:
            extras["batch_curve"] = {str(Bc): {"serial": curve_point(Bc, 0), "pipelined": curve_point(Bc, 1)} for Bc in (1024, 4096, 16384, 32768)}
            extras["batch_curve_note"] = ("proofs/s (prove + verify) with Bc proofs in flight, same table and inputs as the headline; serial = one "
                                          "lane, calls as in the timed region; pipelined = mp_set_pipeline(1): verify of batch k beside prove "
                                          "of batch k + 1, two sets of prover outputs")
        # ---- the same step through the reference-shaped host-buffer API (mp_shuffle_and_remask_batch + mp_verify_shuffle_batch: inputs
        # start in host memory, outputs end there; PCIe-inclusive, never `value`), page-locked (mp_host_alloc) and ordinary buffers
        def api_host(Bh):
            import ctypes
            import numpy as np
            lib = table.lib
            src = {"decks": decks[:Bh].cpu().numpy(), "rho": factors[:Bh].cpu().numpy(), "perms": perms[:Bh].cpu().numpy().astype(np.uint32),
                   "seeds": seeds[:Bh].cpu().numpy()}
            outs = {"od": ((Bh, N * CB), np.uint8), "op": ((Bh, proof_bytes), np.uint8), "st": ((Bh,), np.int32), "sv": ((Bh,), np.int32)}
            res = {}
            for kind in ("pinned", "pageable"):
                held, a, o = [], {}, {}
                if kind == "pinned":
                    def pinned(shape, dtype):
                        nb = int(np.prod(shape)) * np.dtype(dtype).itemsize
                        p_ = lib.mp_host_alloc(nb)
                        assert p_, "mp_host_alloc failed"
                        held.append(p_)
                        return np.frombuffer((ctypes.c_uint8 * nb).from_address(p_), dtype=dtype).reshape(shape)
                    for k_, v in src.items():
                        a[k_] = pinned(v.shape, v.dtype)
                        a[k_][...] = v
                    o = {k_: pinned(*v) for k_, v in outs.items()}
                else:
                    a = {k_: np.ascontiguousarray(v) for k_, v in src.items()}
                    o = {k_: np.empty(*v) for k_, v in outs.items()}
                ptr = lambda x: x.ctypes.data_as(ctypes.c_void_p)

                def run():
                    assert lib.mp_shuffle_and_remask_batch(table.h, Bh, ptr(a["decks"]), ptr(a["rho"]), ptr(a["perms"]), ptr(a["seeds"]),
                                                           ptr(o["od"]), ptr(o["op"]), ptr(o["st"])) == 0
                    assert lib.mp_verify_shuffle_batch(table.h, Bh, ptr(a["decks"]), ptr(o["od"]), ptr(o["op"]), ptr(o["sv"])) == 0
                run()
                reps = 2 if Bh >= 65536 else 6
                t1 = time.perf_counter()
                for _ in range(reps):
                    run()
                res[kind] = round(Bh * reps / (time.perf_counter() - t1), 1)
                assert not o["st"].any() and not o["sv"].any()
                assert bytes(o["op"][Bh // 2]) == bytes(ref_rows[Bh // 2].cpu().numpy().tobytes()), "host-buffer API: different proof bytes"
                del a, o
                for p_ in held:
                    lib.mp_host_free(p_)
            return res
        extras["api_host_value"] = {str(Bh): api_host(Bh) for Bh in sorted({min(B, 262144), min(B, 16384)}, reverse=True)}
        extras["api_host_note"] = ("mp_shuffle_and_remask_batch + mp_verify_shuffle_batch, proofs/s: inputs and outputs in host memory "
                                   "(22 KB per proof over PCIe), chunks of 65 536 proofs pipelined over three streams")

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel (live HIP-event timings of the timed region)
    stats = table.plan_stats()
    # group verification (include/mpshuffle.h: mp_set_group_verify): the verifier's screen of this batch is one bucket-method MSM per group of
    # `gl` proofs -- every point of the group's proofs once, the n + 5 fixed bases once per group -- instead of one Straus MSM per proof
    gl = 0
    if workload in ("pairs", "mixed") and not args.per_equation and args.keyed == 0:
        gl = table.group_size(Bs if workload == "pairs" else B // 2)
    # chain verification (mp_set_chain_group / mp_set_chain_slice): the chains of chain_G tables share one equation, a call's tables go in
    # passes of chain_pass tables (what the free memory holds of the 68 KB of workspace per link in flight)
    chain_G, chain_pass = 1, None
    if workload == "chain32" and not args.per_link_verify:
        chain_pass = table.chain_last_slice() or B
        chain_G = max(1, table.chain_group_size(chain_pass, extras["chain_links_per_equation"], True))
        extras["chain_tables_per_equation"], extras["chain_tables_per_pass"] = chain_G, chain_pass
    SCALAR_BITS = {"stark": 252, "bn254": 254, "secp256k1": 256, "bls12_377": 253}[curve]

    def bucket_geometry(K):
        """window width the engine picks for a bucket-method MSM of K terms (kernels_bucket.hpp bk_bits_for), its windows per scalar,
        buckets per lane and the point additions of the wave-wide reduction of one window"""
        c = 14 if K >= 200000 else (12 if K >= 50000 else (11 if K >= 40000 else (10 if K >= 12000 else (9 if K >= 6000 else 8))))      # (12 and 13: the split pipeline, one reducing wave per window)
        nb = (1 << (c - 1)) // 64
        return {"bits": c, "windows": (SCALAR_BITS + c - 1) // c, "buckets_per_lane": nb, "reduction_adds": 13 + 2 * nb - 3}      # (round 6: min(k, q - k) is recoded: ceil(bits / c) windows)
    per_proof_pts = 4 * N + 11 * m + 8
    geo_own = bucket_geometry(per_proof_pts + 1)          # a proof's own bucket jobs (large decks: the merged equation and the prover's long products)
    geo_grp = None
    if gl:
        geo_grp = bucket_geometry(gl * per_proof_pts)
        # (large decks: the per-proof screen ran the merged equation as a bucket job of its own; the group equation replaces it)
        own_terms = stats.get("bucket_terms", 0) - (per_proof_pts + 1 if stats["verify"].get("var_terms", 0) == 0 and stats.get("bucket_terms", 0) >= per_proof_pts else 0)
        own_jobs = stats.get("bucket_jobs", 0) - (1 if own_terms != stats.get("bucket_terms", 0) else 0)
        stats["verify"] = {"fixed_terms": (n + 5) / gl, "var_terms": 0, "fixed_jobs": 4.0 / gl, "var_jobs": 0, "table_bases": 0, "combine_terms": 5.0 / gl}
        stats["bucket_terms"], stats["bucket_jobs"] = max(own_terms, 0), max(own_jobs, 0)
        stats["group_terms"] = per_proof_pts
        stats["group_jobs"] = 1.0 / gl
        stats["verify_group_size"] = gl
        stats["group_bucket_geometry"] = geo_grp
    dom = max(prof.items(), key=lambda kv: kv[1][1])
    dom_name, (dom_count, dom_ms) = dom
    kernel_ms_total = sum(v[1] for v in prof.values())

    def alg_bytes_per_proof(kernel):
        """ALGORITHMIC bytes one proof needs from this kernel class in one step (prove + verify launches):
        scalars 32 B, points PB, Jacobian results 1.5 PB ("algorithmic bytes")."""
        pv = [stats["prove"], stats["verify"]]
        if kernel == "k_var_msm":
            return sum(s["var_terms"] * (32 + PB) + s["var_jobs"] * 3 * PB // 2 for s in pv)
        if kernel in ("k_bucket_msm", "k_bucket_acc"):
            return (stats.get("bucket_terms", 0) + stats.get("group_terms", 0)) * (32 + PB) + (stats.get("bucket_jobs", 0) + stats.get("group_jobs", 0)) * 3 * PB // 2
        if kernel == "k_fixed_msm":
            return sum(s["fixed_terms"] * 32 + s["fixed_jobs"] * 3 * PB // 2 for s in pv)
        if kernel == "k_table":
            return sum(s["table_bases"] * (PB + 16 * PB) for s in pv)
        if kernel == "k_remask":          # the deck, one masking factor per card, the re-encrypted deck as Jacobian points
            return 2 * N * PB + N * 32 + 2 * N * 3 * PB // 2
        if kernel == "k_normalize":
            return sum(s["table_bases"] * 16 * (3 * PB // 2 + PB) for s in pv)
        return None
    whole_path_bytes = (2 * N * PB + N * 32 + 4 * N + (n + 4) * PB) + (2 * N * PB + proof_bytes) + (4 * N * PB + proof_bytes + (n + 4) * PB)
    per_proof = alg_bytes_per_proof(dom_name) or whole_path_bytes
    # per-launch figures: summed over the launches of the timed region
    dom_bytes = per_proof * total_proofs / world
    achieved_gbs = dom_bytes / (dom_ms * 1e-3) / 1e9
    # multiply-add count of one prove+verify: static plan (stats) x instruction counts of the compiled group law
    mc = load_mad_counts(curve)
    int_mul = None
    if mc:
        vw, fw = stats["var_windows"], stats["fixed_windows"]
        mads_per_proof = 0
        chain_eq = workload == "chain32" and not args.per_link_verify
        for side in ("prove", "verify"):
            st_ = stats[side]
            if side == "verify" and chain_eq:
                # one chain equation per table instead of L per-link equations (engine_core.hpp build_chain_plan): a bucket MSM
                # over the (L+1) decks, the L proofs' points and the key, plus one fixed-base term per shared generator
                # (round 5: the chains of chain_G tables in one equation; a chain longer than --chain-max-links in consecutive sub-chains)
                L_, lc_ = args.players, extras["chain_links_per_equation"]
                nsub_ = (L_ + lc_ - 1) // lc_
                k_terms = (lc_ + 1) * 2 * N + lc_ * (11 * m + 7) + 1
                geo_c = bucket_geometry(chain_G * k_terms)
                bwc = geo_c["windows"]
                fmc = mc["field"]
                red = geo_c["reduction_adds"] * 64 * (12 * fmc["mul"] + 2 * fmc["sqr"]) + geo_c["bits"] * mc["dbl"] + mc["jac"]
                mads_per_proof += nsub_ * (k_terms * bwc * mc["madd"] + (bwc * red + (n + 2) * fw * mc["madd"]) / chain_G) / L_
                continue
            fixed_madds = st_["fixed_terms"] * fw
            if side == "prove":
                fixed_madds -= N * (fw - 1)               # c_A commits pi(i)+1 <= N: one non-zero window per term
                fixed_madds += 2 * N * (fw + 1)           # re-encryption: 2N fixed-base scalar-muls + one addition each
                norm_points = 2 * N + 11 * m + 7          # shuffled deck + proof points
            else:
                norm_points = 0
            mads_per_proof += (fixed_madds + st_["var_terms"] * vw) * mc["madd"]
            mads_per_proof += st_["var_jobs"] * (vw - 1) * 5 * mc["dbl"]
            mads_per_proof += st_["table_bases"] * 15 * mc["aff"]
            mads_per_proof += st_["combine_terms"] * mc["jac"] + norm_points * mc["norm"]
        # Toom-Cook evaluation of the ciphertext polynomials (k_toom_points, kernels_msm.hpp): per vector position and pair +-x a
        # Horner scheme in x^2 over the even and the odd coefficients (Jacobian: doubling 3M+(4+2a)S, mixed addition 8M+3S,
        # addition 11M+5S), multiplications by the small integers x^2 and x as double-and-add
        tm = stats.get("toom_points_m", 0)
        if tm:
            fmt = mc["field"]
            jdbl = 3 * fmt["mul"] + (4 + 2 * a1_of(mc)) * fmt["sqr"]
            jmadd = 8 * fmt["mul"] + 3 * fmt["sqr"]

            def mul_small(k):
                return 0 if k == 1 else (k.bit_length() - 1) * jdbl + (bin(k).count("1") - 1) * mc["jac"]
            top_even, top_odd = (tm - 1) & ~1, (tm - 1) if (tm - 1) & 1 else tm - 2
            per_pos = 0
            for p_ in range(tm - 1):
                x_ = 1 if p_ == 0 else (p_ + 1) // 2 + 1
                steps_ = top_even // 2 + (top_odd - 1) // 2
                per_pos += steps_ * (mul_small(x_ * x_) + jmadd) + mul_small(x_) + 2 * mc["jac"]
            mads_per_proof += 2 * n * per_pos + (2 * tm - 2) * 2 * n * mc["norm"]        # + normalisation of the evaluated vectors
        # bucket-method MSMs: one mixed addition per term and window; per (MSM, window) the wave-wide reduction on 64 lanes
        # (XYZZ + XYZZ, 12M+2S; 13 + 2 NB - 3 of them) and the fold (c doublings + 1 addition)
        bw = geo_own["windows"]
        fm = mc["field"]
        xyzz_add = 12 * fm["mul"] + 2 * fm["sqr"]
        for terms_k, jobs_k, geo in (("bucket_terms", "bucket_jobs", geo_own), ("group_terms", "group_jobs", geo_grp)):
            if geo:
                mads_per_proof += stats.get(terms_k, 0) * geo["windows"] * mc["madd"]
                mads_per_proof += stats.get(jobs_k, 0) * geo["windows"] * (geo["reduction_adds"] * 64 * xyzz_add + geo["bits"] * mc["dbl"] + mc["jac"])
        mads = mads_per_proof * total_proofs / world
        int_mul = {"bound": "v_mad_u64_u32 issue", "achieved": round(mads / (kernel_ms_total * 1e-3) / 1e9, 1),
                   "peak": INT_MAD_PEAK_G, "unit": "Gmad/s",
                   "frac": mads / (kernel_ms_total * 1e-3) / 1e9 / INT_MAD_PEAK_G,
                   "mads_per_proof": int(mads_per_proof), "mads_per_op": {k: v for k, v in mc.items() if k not in ("field", "a1", "issue")},
                   "plan_stats": stats, "bucket_windows": (geo_grp or geo_own)["windows"],
                   "mads_per_field_op": mc["field"],
                   "note": "32x32+64 multiply-adds (v_mad_u64_u32 + v_mad_i64_i32) counted in the gfx950 assembly of THIS build "
                           "(mental-poker_amd/mad_counts.json, tools/gen_mad_counts.py) x static plan, over the kernel time of the WHOLE step; "
                           "peak = 1024 SIMDs x 16 lanes/clk x 2.4 GHz (sustained in tools/microbench/roof.hip: 35.3 T/s at 2.28 GHz and 1.17 kW). "
                           "The multiplies are ~55 % of the issue slots of the group law: the binding figure is roofline.compute"}
    # ---- compute bound of the dominant kernel: VALU issue time.  Wave-level group operations of the kernel (static plan) x the issue
    # cycles of one operation (4 x half-rate + 2 x full-rate VALU instructions of its main path, counted in the gfx950 ISA of this
    # build) / 1024 SIMDs = the cycles every SIMD needs at the very least; divided by the kernel's measured time = the issue rate it
    # sustained, against the shader clock (2.4 GHz maximum; the clock measured under this kernel if a PMC pass of this build exists)
    compute = None
    if mc and mc.get("issue"):
        iss = mc["issue"]
        vw, fw = stats["var_windows"], stats["fixed_windows"]
        ops = None                      # per proof pair (prove + verify launches of one step), lane level
        if dom_name == "k_var_msm":
            # (chain32 with chain verification: the verifier's equation runs on the bucket kernel, only the prover uses k_var_msm)
            sides = ("prove",) if (workload == "chain32" and not args.per_link_verify) else ("prove", "verify")
            ops = {"madd": sum(stats[s_]["var_terms"] for s_ in sides) * vw,
                   "dbl": sum(stats[s_]["var_jobs"] for s_ in sides) * (vw - 1) * 5}
        elif dom_name == "k_fixed_msm":
            ops = {"madd": sum(stats[s_]["fixed_terms"] for s_ in ("prove", "verify")) * fw - N * (fw - 1)}
        elif dom_name == "k_remask":
            ops = {"madd": 2 * N * (fw + 1)}
        elif dom_name in ("k_bucket_msm", "k_bucket_acc") and "xadd" in iss:
            if workload == "chain32" and not args.per_link_verify:
                L_, lc_ = args.players, extras["chain_links_per_equation"]
                nsub_ = (L_ + lc_ - 1) // lc_
                kc_ = (lc_ + 1) * 2 * N + lc_ * (11 * m + 7) + 1
                parts_ = [(nsub_ * kc_ / L_, nsub_ / (chain_G * L_), bucket_geometry(chain_G * kc_))]
            else:
                parts_ = [(stats.get("bucket_terms", 0), stats.get("bucket_jobs", 0), geo_own)]
                if geo_grp:
                    parts_.append((stats["group_terms"], stats["group_jobs"], geo_grp))
            # one mixed addition per term and window (64 lanes share a window's terms evenly at best); per (MSM, window) the wave-wide
            # reduction is 13 + 2 NB - 3 full additions on all 64 lanes
            ops = {"madd": sum(t_ * g_["windows"] for t_, j_, g_ in parts_), "xadd": sum(j_ * g_["windows"] * g_["reduction_adds"] * 64 for t_, j_, g_ in parts_)}
        if ops and sum(ops.values()) > 0:       # (plan_stats describes the throughput plan: small batches on the finer splits have no entry)
            cyc_per_proof = sum(ops[k] * iss[k]["cycles"] for k in ops)                     # lane-level issue cycles x 1 lane
            waves_cycles = cyc_per_proof * (total_proofs / world) / 64.0                        # wave-level instructions issue for 64 lanes at once
            achieved_ghz = waves_cycles / SIMDS / (dom_ms * 1e-3) / 1e9
            # second view, the one the SQ counters see: every VALU instruction takes one 4-cycle issue slot of its SIMD (in streams that
            # mix the two classes the cheap instructions do not get their 2-cycle rate: roof microbenchmark, mad:cheap rows)
            insts_per_proof = sum(ops[k] * (iss[k]["half_rate"] + iss[k]["full_rate"]) for k in ops)
            slot_ginst = insts_per_proof * (total_proofs / world) / 64.0 / (dom_ms * 1e-3) / 1e9
            slots = {"valu_insts_per_op": {k: iss[k]["half_rate"] + iss[k]["full_rate"] for k in ops},
                     "achieved_G_wave_insts_per_s": round(slot_ginst, 2), "peak_G_wave_insts_per_s": SIMDS * CLK_MAX_GHZ / 4.0,
                     "frac": slot_ginst / (SIMDS * CLK_MAX_GHZ / 4.0), "frac_at_measured_clock": None,
                     "note": "one VALU instruction per 4-cycle slot per SIMD; static instruction counts x plan"}
            compute = {"bound": "valu_issue", "kernel": dom_name, "achieved": round(achieved_ghz, 4), "peak": CLK_MAX_GHZ,
                       "unit": "G issue-cycles/s per SIMD (= GHz of fully used VALU issue port)", "frac": achieved_ghz / CLK_MAX_GHZ,
                       "ops_per_proof": {k: round(v, 1) for k, v in ops.items()},
                       "issue_cycles_per_op": {k: iss[k] for k in ops},
                       "clock_mhz": None, "peak_at_measured_clock": None, "frac_at_measured_clock": None, "issue_slots": slots,
                       "note": "ideal issue time = sum over wave-level operations of (4 x half-rate + 2 x full-rate VALU instructions) / 1024 "
                               "SIMDs; instruction counts from the gfx950 ISA of this build (mad_counts.json: valu_issue), classes and "
                               "the no-co-issue rule from tools/microbench/roof.hip (profiles/r03_roof.json)"}
    # HBM traffic of the dominant kernel: PMC pass (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs, gfx950 x2
    # correction applied to FETCH_SIZE) of THIS build at THIS configuration, if one is committed under profiles/
    src_hash = engine_source_hash()
    traffic, traffic_src = None, None
    pmc_path, pmc = find_pmc_summary(src_hash, curve, m, n, B, workload)
    if pmc and dom_name in pmc.get("kernels", {}):
        e_ = pmc["kernels"][dom_name]
        traffic = e_.get("hbm_bytes_per_proof_per_step", e_["hbm_bytes_per_proof_per_step_corrected"]) * (total_proofs / world) / dom_count
        traffic_src = pmc_path
        if compute and e_.get("clock_mhz"):
            # effective shader clock under this kernel = GRBM_GUI_ACTIVE / 8 XCDs / dispatch duration, both of the SAME profiled pass
            compute["clock_mhz"] = round(e_["clock_mhz"], 1)
            compute["peak_at_measured_clock"] = round(e_["clock_mhz"] / 1e3, 4)
            compute["frac_at_measured_clock"] = min(1.0, compute["achieved"] / (e_["clock_mhz"] / 1e3))
            compute["clock_source"] = pmc_path
            compute["issue_slots"]["frac_at_measured_clock"] = min(1.0, compute["issue_slots"]["achieved_G_wave_insts_per_s"] / (SIMDS * e_["clock_mhz"] / 1e3 / 4.0))
        if compute and e_.get("SQ_INSTS_VALU_per_launch"):
            compute["valu_insts_per_launch_pmc"] = e_["SQ_INSTS_VALU_per_launch"]
            # measured, not modelled: SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x shader cycles of the dispatch), same PMC pass
            compute["issue_slots"]["measured_pmc"] = e_.get("valu_slot_utilisation")
    roofline = {
        "bound": "hbm", "kernel": dom_name, "achieved": round(achieved_gbs, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
        "avg_launch_ms": dom_ms / dom_count, "launches": dom_count,
        "alg_bytes_per_launch": dom_bytes / dom_count, "alg_bytes_per_proof": per_proof,
        "note": "path is bound by VALU issue, not by HBM (SURVEY 8d3): see compute (dominant kernel) and int_mul (whole step)",
        "compute": compute,
        "int_mul": int_mul,
        "whole_path_hbm_frac": value / world * whole_path_bytes / 1e9 / HBM_PEAK_GBS,
        "kernels_ms": {k: round(v[1], 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])},
        "kernel_launches": {k: v[0] for k, v in prof.items()}, "priming_launches": priming_launches,
        "engine_src": src_hash,
    }

    # one row per kernel that takes >= 3 % of the step: per-launch time, algorithmic bytes per launch and their fraction of the HBM peak,
    # and -- from the PMC pass of THIS build at THIS configuration, if one is committed under profiles/ -- the VALU issue slots it used,
    # its clock and the bytes the counters saw per launch
    rows_k = []
    for k_, (cnt_, ms_) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
        if ms_ < 0.03 * kernel_ms_total or not cnt_:
            continue
        ab = alg_bytes_per_proof(k_)
        row = {"kernel": k_, "share": round(ms_ / kernel_ms_total, 4), "ms": round(ms_ / cnt_, 3), "launches_per_step": round(cnt_ / args.steps, 2),
               "alg_bytes": None if ab is None else int(ab * total_proofs / world / cnt_),
               "hbm_frac": None if ab is None else round(ab * total_proofs / world / (ms_ * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
               "valu_slots": None, "clock_mhz": None, "pmc_bytes": None}
        e_ = (pmc or {}).get("kernels", {}).get(k_) if pmc else None
        if e_:
            row["valu_slots"] = e_.get("valu_slot_utilisation")
            row["clock_mhz"] = e_.get("clock_mhz")
            pb_ = e_.get("hbm_bytes_per_proof_per_step", e_.get("hbm_bytes_per_proof_per_step_corrected"))
            row["pmc_bytes"] = None if pb_ is None else int(pb_ * (total_proofs / world) / cnt_)
        rows_k.append(row)
    roofline["kernels"] = rows_k
    roofline["kernels_pmc_source"] = pmc_path if pmc else None

    # ---- CPU baseline: the oracle's C++ restatement (port), single thread, bounded sample
    cpu = None
    if not args.no_cpu_baseline:                  # rank 0 only (the other ranks have returned); N > 1 lines carry it too
        import coracle as co
        numa = pin_to_gpu_numa_node(local)        # the host cores next to this rank's GPU
        it = args.cpu_iters if N <= 64 else max(2, args.cpu_iters * 52 // (N * max(1, m // 2)))
        t_p, t_v = co.bench(curve, m, n, 99, it)
        cpu = {"value": it / (t_p + t_v), "unit": "proofs/s", "cores": 1, "kind": "port",
               "sample": "%d prove+verify pairs, %d-card deck (m=%d,n=%d), %s, single thread; prove %.1f ms verify %.1f ms each"
                         % (it, N, m, n, curve, 1e3 * t_p / it, 1e3 * t_v / it),
               "host_cores_available": os.cpu_count(), "numa": numa}
        # the same port on every host core, one independent proof stream per thread (the reference itself is
        # single-threaded: BASELINE.md section 2); ctypes releases the GIL for the duration of the C call
        from concurrent.futures import ThreadPoolExecutor
        T = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        quota = None
        try:                                   # cgroup v2 CPU quota of the container, if any ("max" = none)
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()
            quota = None if q == "max" else float(q) / float(per)
        except (OSError, ValueError):
            pass
        T = max(1, min(T, 64, int(quota + 0.5) if quota else T))    # bounded: the leg must stay within seconds
        it_mt = max(1, min(8, it // 4))
        t0 = time.perf_counter()
        with ThreadPoolExecutor(T) as ex:
            list(ex.map(lambda i: co.bench(curve, m, n, 1000 + i, it_mt), range(T)))
        wall = time.perf_counter() - t0
        cpu["all_cores"] = {"value": T * it_mt / wall, "unit": "proofs/s", "cores": T,
                            "cgroup_cpu_quota": quota,
                            "sample": "%d threads x %d prove+verify pairs, one proof stream per thread, %.1f s wall" % (T, it_mt, wall)}

    # ---- the same device-resident step on a table from plain mp_table_create (the call INTEGRATION.md's Rust `table_for` makes): the
    # engine sizes the fixed-base windows by the HBM that is free.  Last, because the benchmarked table has to go first.
    if workload == "pairs" and world == 1 and not args.no_extras and args.keyed == 0 and not args.per_equation and S == 1 and not args.pipeline:
        table.close()
        del tables[:]
        torch.cuda.empty_cache()
        t_tab = time.perf_counter()
        dtab = eng.table(m, n, params, pk, fb_bits=0)
        eng.sync()
        extras["default_table_build_s"] = round(time.perf_counter() - t_tab, 3)
        extras["default_table_window_bits"] = dtab.fb_bits
        dtab.reserve(B)

        def dstep():
            dtab.shuffle_and_remask_batch_dev(B, decks.data_ptr(), factors.data_ptr(), perms.data_ptr(), seeds.data_ptr(), out_decks.data_ptr(),
                                              out_proofs.data_ptr(), st_p.data_ptr())
            dtab.verify_shuffle_batch_dev(B, decks.data_ptr(), out_decks.data_ptr(), out_proofs.data_ptr(), st_v.data_ptr())
        dstep()
        barrier()
        t1 = time.perf_counter()
        dstep()
        dstep()
        barrier()
        extras["default_table_value"] = round(2 * B / (time.perf_counter() - t1), 1)
        assert int((st_p != 0).sum().item()) + int((st_v != 0).sum().item()) == 0
        dtab.close()

    limbs = {"stark": "9x29-bit lazy base field", "secp256k1": "9x29-bit lazy base field (signed sparse limbs)",
             "bls12_377": "12x32 base field"}.get(curve, "8x32 base field")
    config = {"workload": "%s: %d-card deck, m=%d n=%d, %s curve, shuffle_and_remask + verify_shuffle" % (workload, N, m, n, curve),
              "unit_counted": units,
              "proofs_per_gpu_per_step": proofs_per_step, "streams": S, "fixed_base_window_bits": args.fb_bits,
              "aggregate_keys": (B if workload == "chain32" else (args.keyed if args.keyed else 1)),
              "verification": ("per equation" if args.per_equation else
                               ("screening pass: one equation per group of %d proofs, weights from every proof of the group, on the bucket-method "
                                "kernel (per-equation pass only to name a failure)" % gl) if gl else
                               "merged screening pass per proof (per-equation pass only to name a failure)"),
              # engine choices by batch size (include/mpshuffle.h: mp_set_transcript_lanes; engine_base.hpp: OVERLAP_MAX_BATCH)
              "transcript_lanes": (args.transcript_lanes or (4 if Bs <= 32768 else 1)),
              # (engine_base.hpp OVERLAP_MAX_BATCH: prove launches of up to 32 768 proofs run their first stretch on two streams, and their
              # per-kernel event times then overlap -- kernels_ms / int_mul are not additive there)
              "prover_streams": (2 if {"pairs": Bs, "chain32": B, "mixed": B // 2}[workload] <= 32768 else 1),
              "parallelism": "%d rank(s), proofs sharded, no data-path collective; parameters broadcast once (%s)" % (world, backend),
              "rccl_world": (dist.get_world_size() if world > 1 else 1), "collective_backend": backend if world > 1 else None,
              "rccl_smoke": smoke_log, "pipeline_depth": args.pipeline,
              "subgroup_check": (not args.no_subgroup_check) if curve == "bls12_377" else None,
              "validated_once": bool(args.validated_once),
              "table_build_s": round(table_build_s, 3),
              "hbm_per_rank_gb": hbm_used_gb,
              "per_rank_proofs": [int(r[0]) for r in rows], "per_rank_failed": [int(r[1]) for r in rows],
              "per_rank_seconds": [round(r[2], 4) for r in rows],
              # telemetry of the timed region, per rank (Telemetry above; None = no hwmon file): what explains a scaling shortfall first
              "per_rank_sclk_mhz": [round(r[3], 1) if r[3] >= 0 else None for r in rows],
              "per_rank_power_w": [round(r[4], 1) if r[4] >= 0 else None for r in rows],
              "per_rank_telemetry_samples": [int(r[5]) for r in rows],
              "per_rank_table_build_s": [round(r[6], 3) for r in rows],
              "per_rank_host_enqueue_ms_per_step": [round(1e3 * r[7] / args.steps, 3) for r in rows],
              "parity_vs_oracle": parity}
    config.update(extras)
    # flat scalars of the nested extras (a driver that keeps only scalar fields still sees them)
    for Bh_, v_ in (extras.get("api_host_value") or {}).items():
        tag_ = "" if int(Bh_) == min(B, 262144) else "_%s" % Bh_
        config["api_host_pinned%s_value" % tag_] = v_["pinned"]
        config["api_host_pageable%s_value" % tag_] = v_["pageable"]
    for Bc_, v_ in (extras.get("batch_curve") or {}).items():
        config["batch_%s_serial" % Bc_] = v_["serial"]
        config["batch_%s_pipelined" % Bc_] = v_["pipelined"]
    for r_ in rows_k[:4]:
        config["%s_ms" % r_["kernel"]] = r_["ms"]
        if r_["valu_slots"] is not None:
            config["%s_valu_slots" % r_["kernel"]] = r_["valu_slots"]
    if digests is not None:
        config["digests"] = digests
    # the scalars a reader wants first come first (a driver that keeps only the head of `config` -- round 5's kept 21 keys -- still has them)
    head = ["workload", "proofs_per_gpu_per_step", "parity_vs_oracle", "one_bad_value", "pct1_bad_value", "pct1_bad_first_value",
            "per_proof_screen_value", "per_equation_value", "api_host_pinned_value", "batch_1024_serial", "batch_1024_pipelined",
            "batch_4096_serial", "batch_16384_serial", "batch_32768_serial", "keyed_value", "hbm_per_rank_gb", "table_build_s",
            "per_rank_sclk_mhz", "per_rank_power_w", "rccl_world", "fixed_base_window_bits", "verification"]
    config = dict([(k, config[k]) for k in head if k in config] + [(k, v) for k, v in config.items() if k not in head])
    out = {
        "metric": "shuffle proofs/sec (prove+verify)", "value": value, "unit": "proofs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "u32 limbs (256-bit Montgomery: %s, 8x32 scalar field)" % limbs,
        "data": "synthetic",
        "config": config,
        "roofline": roofline, "cpu_baseline": cpu,
    }
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
