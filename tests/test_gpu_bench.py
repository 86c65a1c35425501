"""GPU tests of bench.py itself (run on the MI355X box): `--gpus N` started from a plain shell launches N ranks on its own,
the sharded run produces byte-identical outputs to an unsharded run of the same proofs, and the JSON line carries what the
driver reads.  The box has one GPU: MP_BENCH_FORCE_DEVICE / MP_BENCH_BACKEND=gloo put both ranks on it and run the
once-per-session collectives over gloo (on an 8-GPU node the same code path uses RCCL)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def run_bench(*args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), env=e, cwd=ROOT,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints ONE json line"
    return json.loads(lines[0])


def _telemetry_keys(c, world):
    """round 6: every rank reports its shader clock and board power over the timed region, its table build and the
    host time its calls took -- what explains a scaling shortfall first"""
    for k in ("per_rank_sclk_mhz", "per_rank_power_w", "per_rank_telemetry_samples", "per_rank_table_build_s", "per_rank_host_enqueue_ms_per_step"):
        assert len(c[k]) == world, k
    assert all(v > 0 for v in c["per_rank_table_build_s"]) and all(v > 0 for v in c["per_rank_host_enqueue_ms_per_step"])
    for f, p_, n_ in zip(c["per_rank_sclk_mhz"], c["per_rank_power_w"], c["per_rank_telemetry_samples"]):
        assert (f is None or 100 < f < 3000) and (p_ is None or 10 < p_ < 2000) and n_ >= 0


COMMON = ["--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-extras", "--fb-bits", "8", "--digest", "--seed-block", "192"]


def test_self_launch_two_ranks_equals_unsharded():
    sharded = run_bench("--gpus", "2", "--batch", "192", *COMMON, env={"MP_BENCH_FORCE_DEVICE": "0", "MP_BENCH_BACKEND": "gloo"})
    assert sharded["n_gpus"] == 2
    assert sharded["config"]["per_rank_proofs"] == [192, 192] and sharded["config"]["per_rank_failed"] == [0, 0]
    assert len(sharded["config"]["per_rank_seconds"]) == 2
    _telemetry_keys(sharded["config"], 2)
    assert abs(sharded["value"] - 384 / (sharded["ms_per_step"] * 1e-3)) < 1e-6 * sharded["value"]
    single = run_bench("--gpus", "1", "--batch", "384", *COMMON)
    assert single["n_gpus"] == 1
    flat = [d for per_rank in sharded["config"]["digests"] for d in per_rank]
    assert flat == single["config"]["digests"][0] and len(flat) == 2 and flat[0] != flat[1]
    assert single["config"]["parity_vs_oracle"] is True and sharded["config"]["parity_vs_oracle"] is True


def test_strong_scaling_two_ranks_equals_unsharded():
    """--scaling strong: 384 proofs in total, 192 per rank; digests equal those of the unsharded run of the same 384 proofs; the line says
    "strong" and carries the timings of the collectives checked before any table was built (tools/rccl_smoke.py)"""
    strong = run_bench("--gpus", "2", "--scaling", "strong", "--batch", "384", *COMMON, env={"MP_BENCH_FORCE_DEVICE": "0", "MP_BENCH_BACKEND": "gloo"})
    assert strong["n_gpus"] == 2 and strong["scaling"] == "strong"
    assert strong["config"]["per_rank_proofs"] == [192, 192] and strong["config"]["per_rank_failed"] == [0, 0]
    assert strong["config"]["rccl_smoke"]["rccl_world"] == 2 and strong["config"]["rccl_smoke"]["broadcast_parameters_ms"] >= 0
    single = run_bench("--gpus", "1", "--batch", "384", *COMMON)
    assert single["scaling"] == "weak" and single["config"]["rccl_smoke"] is None
    flat = [d for per_rank in strong["config"]["digests"] for d in per_rank]
    assert flat == single["config"]["digests"][0] and len(flat) == 2


def test_json_line_contract_and_extras():
    d = run_bench("--batch", "16384", "--steps", "2", "--warmup", "1", "--fb-bits", "8", "--cpu-iters", "2")      # (throughput plan: > 14 336 proofs)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-6
    assert r["traffic"] is None or r["traffic_source"].startswith("profiles/")      # only from a PMC pass of THIS build and batch
    assert r["int_mul"]["mads_per_op"]["madd"] == 900 and r["int_mul"]["mads_per_op"]["dbl"] == 828   # STARK, counted in the assembly
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 1
    c = r["compute"]                                            # VALU-issue bound of the dominant kernel: a fraction, never above 1
    assert c["bound"] == "valu_issue" and c["kernel"] == r["kernel"] and 0 < c["frac"] <= 1 and c["peak"] == 2.4
    assert c["issue_cycles_per_op"]["madd"]["cycles"] == 4 * c["issue_cycles_per_op"]["madd"]["half_rate"] + 2 * c["issue_cycles_per_op"]["madd"]["full_rate"]
    assert c["frac_at_measured_clock"] is None or c["frac_at_measured_clock"] <= 1
    assert d["config"]["rccl_world"] == 1 and d["config"]["table_build_s"] > 0 and d["config"]["hbm_per_rank_gb"] > 0
    assert d["config"]["per_equation_value"] > 0 and d["config"]["keyed_value"] > 0
    # round 4: the reference-shaped calls beside the headline (host-buffer API incl. PCIe; a table from plain mp_table_create)
    api = d["config"]["api_host_value"]["16384"]
    assert api["pinned"] > 0 and api["pageable"] > 0
    assert d["config"]["default_table_value"] > 0 and d["config"]["default_table_window_bits"] in (16, 20, 21)
    # round 5: what rejection costs, as flat scalars beside the headline; one row per kernel >= 3 % of the step; flat copies of the nested extras
    c = d["config"]
    assert 0 < c["pct1_bad_value"] <= c["one_bad_value"] * 1.05 and c["one_bad_reverified"] >= 1 and c["pct1_bad_first_value"] > 0
    assert c["api_host_pinned_value"] == api["pinned"] and c["api_host_pageable_value"] == api["pageable"]
    rows = r["kernels"]
    assert rows and rows[0]["kernel"] == r["kernel"] and all(row["share"] >= 0.03 and row["ms"] > 0 for row in rows)
    assert abs(sum(row["share"] for row in rows) - 1.0) < 0.35 and "%s_ms" % rows[0]["kernel"] in c


def test_batch_curve_on_the_line():
    """config.batch_curve: proofs/s with 1 024 .. 32 768 proofs in flight, serial and with the verify calls pipelined, measured after the
    timed region on the benchmarked table"""
    d = run_bench("--batch", "32768", "--steps", "1", "--warmup", "1", "--fb-bits", "16", "--no-cpu-baseline")
    bc = d["config"]["batch_curve"]
    assert sorted(bc, key=int) == ["1024", "4096", "16384", "32768"]
    for k, v in bc.items():
        assert v["serial"] > 0 and v["pipelined"] > 0
    assert bc["32768"]["serial"] > bc["1024"]["serial"]


@pytest.mark.parametrize("workload,extra", [("chain32", ["--batch", "256", "--players", "4"]),
                                            ("chain32", ["--batch", "256", "--players", "4", "--per-link-verify"]),
                                            ("chain32", ["--batch", "256", "--players", "4", "--chain-slice", "96", "--chain-group", "4"]),
                                            ("mixed", ["--batch", "512"])])
def test_workload_modes(workload, extra):
    d = run_bench("--workload", workload, "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--fb-bits", "8", *extra)
    assert d["config"]["workload"].startswith(workload) and d["config"]["parity_vs_oracle"] is True and d["value"] > 0
    if workload == "mixed":
        assert "secp256k1" in d["config"]["workload"]
    if "--chain-group" in extra:      # (round 5) passes of 96, 96 and 64 tables, the chains of 4 tables per equation
        assert d["config"]["chain_tables_per_equation"] == 4 and d["config"]["chain_tables_per_pass"] == 96


@pytest.mark.parametrize("workload,extra,per_rank", [("chain32", ["--batch", "96", "--players", "32"], 96 * 32), ("mixed", ["--batch", "384"], 192)])
def test_two_ranks_other_workloads(workload, extra, per_rank):
    """config 3 (32 dependent shuffles per table, one chain equation per table) and config 5 (mixed prove / verify stream on secp256k1)
    through the N > 1 path: two self-launched ranks, every proof accepted on both, the N > 1 line keeps cpu_baseline and roofline"""
    d = run_bench("--gpus", "2", "--workload", workload, "--steps", "1", "--warmup", "1", "--fb-bits", "8", "--cpu-iters", "2", *extra,
                  env={"MP_BENCH_FORCE_DEVICE": "0", "MP_BENCH_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["config"]["rccl_world"] == 2 and d["config"]["collective_backend"] == "gloo"
    assert d["config"]["per_rank_proofs"] == [per_rank, per_rank] and d["config"]["per_rank_failed"] == [0, 0]
    assert d["config"]["parity_vs_oracle"] is True
    assert d["cpu_baseline"] is not None and d["cpu_baseline"]["kind"] == "port" and d["roofline"]["kernel"].startswith("k_")


def test_eight_ranks_weak_and_strong_equal_unsharded():
    """the world size the driver will use: 8 self-launched ranks (all on the one GPU of the test box, collectives over gloo), weak
    (192 proofs per rank) and strong (1 536 in total); every rank reports, every proof is accepted, and the concatenated digests are
    those of ONE unsharded run of the same 1 536 proofs"""
    env = {"MP_BENCH_FORCE_DEVICE": "0", "MP_BENCH_BACKEND": "gloo"}
    weak = run_bench("--gpus", "8", "--batch", "192", *COMMON, env=env)
    strong = run_bench("--gpus", "8", "--scaling", "strong", "--batch", "1536", *COMMON, env=env)
    single = run_bench("--gpus", "1", "--batch", "1536", *COMMON)
    for d, scaling in ((weak, "weak"), (strong, "strong")):
        c = d["config"]
        assert d["n_gpus"] == 8 and d["scaling"] == scaling and c["rccl_world"] == 8 and c["collective_backend"] == "gloo"
        assert c["per_rank_proofs"] == [192] * 8 and c["per_rank_failed"] == [0] * 8 and len(c["per_rank_seconds"]) == 8
        _telemetry_keys(c, 8)
        assert abs(d["value"] - 1536 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
        assert c["rccl_smoke"]["rccl_world"] == 8 and c["parity_vs_oracle"] is True
        flat = [x for per_rank in c["digests"] for x in per_rank]
        assert flat == single["config"]["digests"][0] and len(flat) == 8 and len(set(flat)) == 8
