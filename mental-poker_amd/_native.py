"""ctypes binding of libmpshuffle.so (include/mpshuffle.h).  Plumbing only: bytes in, bytes out.

The library is the HIP engine; there is no CPU path.  `load()` raises if the shared object is missing
and `Engine(...)` raises `NoDeviceError` if no MI355X is visible.
"""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libmpshuffle.so")
ROOT = os.path.dirname(HERE)

CURVE_IDS = {"stark": 0, "bn254": 1, "secp256k1": 2, "bls12_377": 3}
MP_ERR_BAD_ENCODING, MP_ERR_BAD_PERMUTATION, MP_ERR_BAD_ARGUMENT, MP_ERR_NO_DEVICE, MP_ERR_INTERNAL = -1, -2, -3, -4, -5

SYMBOLS = [
    "mp_ctx_create", "mp_ctx_destroy", "mp_last_error", "mp_check_name", "mp_proof_size", "mp_params_size",
    "mp_point_size", "mp_proof_size_curve", "mp_params_size_curve", "mp_set_merged_verify", "mp_set_subgroup_check", "mp_set_bucket_min", "mp_set_bucket_bits", "mp_set_bucket_split", "mp_set_validated", "mp_deck_validate_dev", "mp_set_chain_max_links", "mp_set_chain_group", "mp_set_chain_slice", "mp_chain_group_size", "mp_chain_last_slice", "mp_set_transcript_lanes", "mp_set_group_lanes", "mp_set_work_split", "mp_set_group_verify", "mp_group_size", "mp_set_group_refine", "mp_reverified_count", "mp_set_group_adapt", "mp_set_pipeline", "mp_set_plan_params", "mp_set_plan_thresholds", "mp_set_toom_cook", "mp_host_alloc", "mp_host_free", "mp_set_io_chunk", "mp_shuffle_and_remask_batch_keys", "mp_table_create_params",
    "mp_verify_shuffle_batch_keys", "mp_shuffle_and_remask_batch_keys_dev", "mp_verify_shuffle_batch_keys_dev",
    "mp_keyset_create", "mp_keyset_destroy", "mp_keyset_size", "mp_shuffle_and_remask_batch_keyset_dev", "mp_verify_shuffle_batch_keyset_dev",
    "mp_setup", "mp_table_create", "mp_table_create_ex", "mp_table_window_bits", "mp_table_destroy", "mp_shuffle_and_remask", "mp_verify_shuffle",
    "mp_shuffle_and_remask_batch", "mp_verify_shuffle_batch", "mp_shuffle_and_remask_batch_dev",
    "mp_verify_shuffle_batch_dev", "mp_verify_shuffle_chain", "mp_verify_shuffle_chain_dev", "mp_sync", "mp_reserve", "mp_set_latency_batch", "mp_remask_batch", "mp_msm", "mp_commit_batch",
    "mp_profile_enable", "mp_profile_report", "mp_work_census", "mp_plan_stats", "mp_sigma_prove_batch",
    "mp_sigma_verify_batch", "mp_blake2s",
    "mp_serialized_point_size", "mp_serialized_deck_size", "mp_serialized_params_size", "mp_serialized_proof_size",
    "mp_points_serialize", "mp_points_deserialize", "mp_deck_serialize", "mp_deck_deserialize", "mp_params_serialize",
    "mp_params_deserialize", "mp_proof_serialize", "mp_proof_deserialize", "mp_points_deserialize_dev", "mp_deck_deserialize_dev",
]


class NativeError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("mpshuffle error %d: %s" % (code, text))
        self.code = code


class NoDeviceError(NativeError):
    pass


SOURCES = ["capi.hip"] + [f % c for c in ("stark", "bn254", "secp256k1", "bls12_377") for f in ("curve_%s.hip", "curve_%s_msm.hip")]


def build(verbose=False):
    """compile the HIP engine for gfx950 in-tree -> mental-poker_amd/libmpshuffle.so
    (one translation unit per curve, compiled in parallel, objects cached under csrc/_obj)"""
    from concurrent.futures import ThreadPoolExecutor
    csrc = os.path.join(HERE, "csrc")
    objdir = os.path.join(csrc, "_obj")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".hpp")]
    headers.append(os.path.join(ROOT, "include", "mpshuffle.h"))
    hdr_time = max(os.path.getmtime(h) for h in headers)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # max-ilp scheduling: +0.8 % on the MSM kernels in a back-to-back A/B (interleaves the mad chains with the carry handling)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-sched-strategy=max-ilp",
             "-mllvm", "-pragma-unroll-threshold=1000000"]   # the 14-limb products (27 columns x 14) must unroll completely: rolled, their operands are re-read from the stack with dynamic indices

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        spath = os.path.join(csrc, src)
        if os.path.exists(obj) and os.path.getmtime(obj) >= max(hdr_time, os.path.getmtime(spath)):
            return obj, False
        cmd = [hipcc] + flags + ["-c", spath, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj, True

    def compile_check():
        """tools/quadcheck/quad_check: on-device unit test of the four-lane group law (kernels_quad.hpp) against the one-lane one;
        built here so that it travels to the GPU box with the library (tests/test_gpu_parity.py runs it).  A test tool: its failure to
        build is reported, never the library's"""
        src = os.path.join(ROOT, "tools", "quadcheck", "quad_check.hip")
        exe = os.path.join(ROOT, "tools", "quadcheck", "quad_check")
        if not os.path.exists(src) or (os.path.exists(exe) and os.path.getmtime(exe) >= max(hdr_time, os.path.getmtime(src))):
            return
        cmd = [hipcc] + [f for f in flags if f != "-fPIC"] + ["-I", csrc, src, "-o", exe]
        if verbose:
            print(" ".join(cmd), flush=True)
        try:
            subprocess.check_call(cmd)
        except (subprocess.CalledProcessError, OSError) as e:
            print("warning: tools/quadcheck/quad_check did not build (%s); the library is unaffected" % e, flush=True)

    with ThreadPoolExecutor(max_workers=len(SOURCES) + 1) as ex:
        chk = ex.submit(compile_check)
        res = list(ex.map(compile_one, SOURCES))
        chk.result()
    objs = [r[0] for r in res]
    if any(r[1] for r in res) or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(o) for o in objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", LIB_PATH] + objs      # serialize_host.hpp uses std::thread
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    # multiply-add counts of the compiled field arithmetic (bench.py's int_mul roofline): regenerated with the library
    mc = os.path.join(HERE, "mad_counts.json")
    gen = os.path.join(ROOT, "tools", "gen_mad_counts.py")
    if os.path.exists(gen) and (not os.path.exists(mc) or os.path.getmtime(mc) < max(hdr_time, os.path.getmtime(gen))):
        import importlib.util
        spec = importlib.util.spec_from_file_location("gen_mad_counts", gen)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.main([f for f in flags if f != "-fPIC"])
    return LIB_PATH


def bind(cdll):
    """declare prototypes on an already opened library"""
    c = ctypes
    u8p, u32p, i32p = c.c_void_p, c.c_void_p, c.c_void_p
    cdll.mp_ctx_create.argtypes = [c.c_int, c.c_int, c.POINTER(c.c_void_p)]
    cdll.mp_ctx_destroy.argtypes = [c.c_void_p]
    cdll.mp_ctx_destroy.restype = None
    cdll.mp_last_error.restype = c.c_char_p
    cdll.mp_check_name.argtypes = [c.c_int]
    cdll.mp_check_name.restype = c.c_char_p
    cdll.mp_proof_size.argtypes = [c.c_uint32, c.c_uint32]
    cdll.mp_proof_size.restype = c.c_size_t
    cdll.mp_params_size.argtypes = [c.c_uint32]
    cdll.mp_params_size.restype = c.c_size_t
    cdll.mp_set_merged_verify.argtypes = [c.c_void_p, c.c_int]
    cdll.mp_set_subgroup_check.argtypes = [c.c_void_p, c.c_int]
    cdll.mp_set_bucket_min.argtypes = [c.c_void_p, c.c_size_t]
    cdll.mp_set_bucket_bits.argtypes = [c.c_void_p, c.c_uint32]
    cdll.mp_set_bucket_split.argtypes = [c.c_void_p, c.c_uint32]
    cdll.mp_set_validated.argtypes = [c.c_void_p, c.c_uint32]
    cdll.mp_deck_validate_dev.argtypes = [c.c_void_p, c.c_size_t, c.c_void_p, c.c_void_p]
    cdll.mp_set_toom_cook.argtypes = [c.c_void_p, c.c_int]
    cdll.mp_set_chain_max_links.argtypes = [c.c_void_p, c.c_uint32]
    cdll.mp_set_chain_group.argtypes = [c.c_void_p, c.c_uint32]
    cdll.mp_set_chain_slice.argtypes = [c.c_void_p, c.c_size_t]
    cdll.mp_chain_group_size.argtypes = [c.c_void_p, c.c_size_t, c.c_uint32, c.c_int]
    cdll.mp_chain_group_size.restype = c.c_uint32
    cdll.mp_chain_last_slice.argtypes = [c.c_void_p]
    cdll.mp_chain_last_slice.restype = c.c_size_t
    cdll.mp_set_transcript_lanes.argtypes = [c.c_void_p, c.c_uint32]
    cdll.mp_set_group_lanes.argtypes = [c.c_void_p, c.c_uint32]
    cdll.mp_set_work_split.argtypes = [c.c_void_p, c.c_int]
    cdll.mp_set_group_verify.argtypes = [c.c_void_p, c.c_uint32, c.c_size_t]
    cdll.mp_group_size.argtypes = [c.c_void_p, c.c_size_t]
    cdll.mp_group_size.restype = c.c_uint32
    cdll.mp_set_group_refine.argtypes = [c.c_void_p, c.c_uint32, c.c_uint32]
    cdll.mp_reverified_count.argtypes = [c.c_void_p]
    cdll.mp_set_group_adapt.argtypes = [c.c_void_p, c.c_int]
    cdll.mp_reverified_count.restype = c.c_uint64
    cdll.mp_set_pipeline.argtypes = [c.c_void_p, c.c_int]
    cdll.mp_set_plan_params.argtypes = [c.c_void_p, c.c_int] + [c.c_uint32] * 5
    cdll.mp_set_plan_thresholds.argtypes = [c.c_void_p] + [c.c_size_t] * 5
    cdll.mp_set_io_chunk.argtypes = [c.c_void_p, c.c_size_t]
    cdll.mp_host_alloc.argtypes = [c.c_size_t]
    cdll.mp_host_alloc.restype = c.c_void_p
    cdll.mp_host_free.argtypes = [c.c_void_p]
    cdll.mp_host_free.restype = None
    cdll.mp_point_size.argtypes = [c.c_int]
    cdll.mp_point_size.restype = c.c_size_t
    cdll.mp_proof_size_curve.argtypes = [c.c_int, c.c_uint32, c.c_uint32]
    cdll.mp_proof_size_curve.restype = c.c_size_t
    cdll.mp_params_size_curve.argtypes = [c.c_int, c.c_uint32]
    cdll.mp_params_size_curve.restype = c.c_size_t
    cdll.mp_setup.argtypes = [c.c_void_p, c.c_uint32, c.c_uint32, u8p, u8p]
    cdll.mp_table_create.argtypes = [c.c_void_p, c.c_uint32, c.c_uint32, u8p, u8p, c.POINTER(c.c_void_p)]
    cdll.mp_table_create_ex.argtypes = [c.c_void_p, c.c_uint32, c.c_uint32, u8p, u8p, c.c_uint32, c.POINTER(c.c_void_p)]
    cdll.mp_table_window_bits.argtypes = [c.c_void_p]
    cdll.mp_table_window_bits.restype = c.c_uint32
    cdll.mp_table_destroy.argtypes = [c.c_void_p]
    cdll.mp_table_destroy.restype = None
    cdll.mp_shuffle_and_remask.argtypes = [c.c_void_p, u8p, u8p, u32p, u8p, u8p, u8p]
    cdll.mp_verify_shuffle.argtypes = [c.c_void_p, u8p, u8p, u8p, c.c_size_t]
    cdll.mp_shuffle_and_remask_batch.argtypes = [c.c_void_p, c.c_size_t, u8p, u8p, u32p, u8p, u8p, u8p, i32p]
    cdll.mp_verify_shuffle_batch.argtypes = [c.c_void_p, c.c_size_t, u8p, u8p, u8p, i32p]
    cdll.mp_shuffle_and_remask_batch_dev.argtypes = [c.c_void_p, c.c_size_t] + [c.c_void_p] * 7
    cdll.mp_verify_shuffle_batch_dev.argtypes = [c.c_void_p, c.c_size_t] + [c.c_void_p] * 4
    cdll.mp_table_create_params.argtypes = [c.c_void_p, c.c_uint32, c.c_uint32, u8p, c.c_uint32, c.POINTER(c.c_void_p)]
    cdll.mp_shuffle_and_remask_batch_keys.argtypes = [c.c_void_p, c.c_size_t, u8p, u8p, u8p, u32p, u8p, u8p, u8p, i32p]
    cdll.mp_verify_shuffle_batch_keys.argtypes = [c.c_void_p, c.c_size_t, u8p, u8p, u8p, u8p, i32p]
    cdll.mp_shuffle_and_remask_batch_keys_dev.argtypes = [c.c_void_p, c.c_size_t] + [c.c_void_p] * 8
    cdll.mp_verify_shuffle_batch_keys_dev.argtypes = [c.c_void_p, c.c_size_t] + [c.c_void_p] * 5
    cdll.mp_keyset_create.argtypes = [c.c_void_p, c.c_size_t, u8p, c.POINTER(c.c_void_p)]
    cdll.mp_keyset_destroy.argtypes = [c.c_void_p]
    cdll.mp_keyset_destroy.restype = None
    cdll.mp_keyset_size.argtypes = [c.c_void_p]
    cdll.mp_keyset_size.restype = c.c_size_t
    cdll.mp_shuffle_and_remask_batch_keyset_dev.argtypes = [c.c_void_p, c.c_void_p, c.c_size_t] + [c.c_void_p] * 8
    cdll.mp_verify_shuffle_batch_keyset_dev.argtypes = [c.c_void_p, c.c_void_p, c.c_size_t] + [c.c_void_p] * 5
    cdll.mp_verify_shuffle_chain.argtypes = [c.c_void_p, c.c_size_t, c.c_uint32, u8p, u8p, u8p, i32p]
    cdll.mp_verify_shuffle_chain_dev.argtypes = [c.c_void_p, c.c_size_t, c.c_uint32] + [c.c_void_p] * 4
    cdll.mp_sync.argtypes = [c.c_void_p]
    cdll.mp_reserve.argtypes = [c.c_void_p, c.c_size_t]
    cdll.mp_set_latency_batch.argtypes = [c.c_void_p, c.c_size_t]
    cdll.mp_remask_batch.argtypes = [c.c_void_p, c.c_size_t, u8p, u8p, u8p]
    cdll.mp_msm.argtypes = [c.c_void_p, c.c_size_t, c.c_size_t, u8p, u8p, u8p]
    cdll.mp_commit_batch.argtypes = [c.c_void_p, c.c_size_t, c.c_size_t, u8p, u8p, u8p]
    cdll.mp_profile_enable.argtypes = [c.c_void_p, c.c_int]
    cdll.mp_profile_report.argtypes = [c.c_void_p, c.c_char_p, c.c_size_t]
    cdll.mp_work_census.argtypes = [c.c_void_p] + [c.POINTER(c.c_uint64)] * 4
    cdll.mp_plan_stats.argtypes = [c.c_void_p, c.POINTER(c.c_uint64)]
    cdll.mp_sigma_prove_batch.argtypes = [c.c_void_p, c.c_size_t, c.c_uint32, u8p, u8p, u8p, u8p, u8p, u8p, i32p]
    cdll.mp_sigma_verify_batch.argtypes = [c.c_void_p, c.c_size_t, c.c_uint32, u8p, u8p, u8p, u8p, i32p]
    cdll.mp_blake2s.argtypes = [u8p, c.c_size_t, u8p]
    for fn, at in (("mp_serialized_point_size", [c.c_int]), ("mp_serialized_deck_size", [c.c_int, c.c_size_t]),
                   ("mp_serialized_params_size", [c.c_int, c.c_uint32]), ("mp_serialized_proof_size", [c.c_int, c.c_uint32, c.c_uint32])):
        getattr(cdll, fn).argtypes = at
        getattr(cdll, fn).restype = c.c_size_t
    cdll.mp_points_serialize.argtypes = [c.c_int, c.c_size_t, u8p, u8p]
    cdll.mp_points_deserialize.argtypes = [c.c_int, c.c_size_t, u8p, u8p]
    cdll.mp_deck_serialize.argtypes = [c.c_int, c.c_size_t, u8p, u8p]
    cdll.mp_deck_deserialize.argtypes = [c.c_int, u8p, c.c_size_t, c.c_size_t, u8p, c.POINTER(c.c_size_t)]
    cdll.mp_params_serialize.argtypes = [c.c_int, c.c_uint32, c.c_uint32, u8p, u8p]
    cdll.mp_params_deserialize.argtypes = [c.c_int, u8p, c.c_size_t, c.c_size_t, c.POINTER(c.c_uint32), c.POINTER(c.c_uint32), u8p]
    cdll.mp_points_deserialize_dev.argtypes = [c.c_void_p, c.c_size_t, c.c_void_p, c.c_void_p, c.c_void_p]
    cdll.mp_deck_deserialize_dev.argtypes = [c.c_void_p, c.c_size_t, c.c_size_t, c.c_void_p, c.c_void_p, c.c_void_p]
    cdll.mp_proof_serialize.argtypes = [c.c_int, c.c_uint32, c.c_uint32, u8p, u8p]
    cdll.mp_proof_deserialize.argtypes = [c.c_int, c.c_uint32, c.c_uint32, u8p, c.c_size_t, u8p]
    return cdll


_LIB = None


def load():
    """open libmpshuffle.so (raises if it was not built: there is no fallback)"""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libmpshuffle.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'`); "
                              "the engine has no CPU fallback")
        _LIB = bind(ctypes.CDLL(LIB_PATH))
    return _LIB


def _in(b):
    b = bytes(b)
    return (ctypes.c_uint8 * max(len(b), 1)).from_buffer_copy(b if b else b"\0")


class Serializer:
    """include/mpshuffle.h "canonical serialisation": arkworks-0.3 compressed bytes <-> wire v1.  Host work in libmpshuffle.so;
    needs no GPU and no context."""

    def __init__(self, curve, lib=None):
        self.lib, self.cid, self.curve = (lib if lib is not None else load()), CURVE_IDS[curve], curve
        self.pb = self.lib.mp_point_size(self.cid)
        self.cb = self.lib.mp_serialized_point_size(self.cid)

    def _chk(self, rc):
        if rc != 0:
            raise NativeError(rc, self.lib.mp_last_error().decode())

    def proof_serialized_size(self, m, n):
        return self.lib.mp_serialized_proof_size(self.cid, m, n)

    def points_serialize(self, wire):
        k = len(wire) // self.pb
        if k * self.pb != len(wire):
            raise NativeError(MP_ERR_BAD_ARGUMENT, "whole wire points expected")
        out = (ctypes.c_uint8 * max(k * self.cb, 1))()
        self._chk(self.lib.mp_points_serialize(self.cid, k, _in(wire), out))
        return bytes(out)[:k * self.cb]

    def points_deserialize(self, data):
        k = len(data) // self.cb
        if k * self.cb != len(data):
            raise NativeError(MP_ERR_BAD_ENCODING, "whole compressed points expected")
        out = (ctypes.c_uint8 * max(k * self.pb, 1))()
        self._chk(self.lib.mp_points_deserialize(self.cid, k, _in(data), out))
        return bytes(out)[:k * self.pb]

    def deck_serialize(self, wire):
        k = len(wire) // (2 * self.pb)
        if 2 * k * self.pb != len(wire):
            raise NativeError(MP_ERR_BAD_ARGUMENT, "whole cards expected")
        out = (ctypes.c_uint8 * self.lib.mp_serialized_deck_size(self.cid, k))()
        self._chk(self.lib.mp_deck_serialize(self.cid, k, _in(wire), out))
        return bytes(out)

    def deck_deserialize(self, data):
        cap = max(len(data) // (2 * self.cb), 1)
        out = (ctypes.c_uint8 * (cap * 2 * self.pb))()
        k = ctypes.c_size_t()
        self._chk(self.lib.mp_deck_deserialize(self.cid, _in(data), len(data), cap, out, ctypes.byref(k)))
        return bytes(out)[:k.value * 2 * self.pb]

    def params_serialize(self, m, n, raw):
        if len(raw) != self.pb * (n + 3):
            raise NativeError(MP_ERR_BAD_ARGUMENT, "parameters: wrong length")
        out = (ctypes.c_uint8 * self.lib.mp_serialized_params_size(self.cid, n))()
        self._chk(self.lib.mp_params_serialize(self.cid, m, n, _in(raw), out))
        return bytes(out)

    def params_deserialize(self, data):
        cap = max(len(data) // self.cb, 1)
        out = (ctypes.c_uint8 * (self.pb * (cap + 3)))()
        m, n = ctypes.c_uint32(), ctypes.c_uint32()
        self._chk(self.lib.mp_params_deserialize(self.cid, _in(data), len(data), cap, ctypes.byref(m), ctypes.byref(n), out))
        return m.value, n.value, bytes(out)[:self.pb * (n.value + 3)]

    def proof_serialize(self, m, n, wire):
        if len(wire) != self.lib.mp_proof_size_curve(self.cid, m, n):
            raise NativeError(MP_ERR_BAD_ARGUMENT, "proof: wrong length")
        out = (ctypes.c_uint8 * self.proof_serialized_size(m, n))()
        self._chk(self.lib.mp_proof_serialize(self.cid, m, n, _in(wire), out))
        return bytes(out)

    def proof_deserialize(self, m, n, data):
        out = (ctypes.c_uint8 * self.lib.mp_proof_size_curve(self.cid, m, n))()
        self._chk(self.lib.mp_proof_deserialize(self.cid, m, n, _in(data), len(data), out))
        return bytes(out)


class Engine:
    """one mp_ctx (GPU + stream + curve)"""

    def __init__(self, curve="stark", device=0, lib=None):
        self.lib = lib if lib is not None else load()
        self.curve = curve
        h = ctypes.c_void_p()
        rc = self.lib.mp_ctx_create(CURVE_IDS[curve], device, ctypes.byref(h))
        if rc != 0:
            text = self.lib.mp_last_error().decode()
            raise (NoDeviceError if rc == MP_ERR_NO_DEVICE else NativeError)(rc, text)
        self.h = h
        self.point_bytes = self.lib.mp_point_size(CURVE_IDS[curve])     # 64; 96 on bls12_377

    def close(self):
        if getattr(self, "h", None):
            self.lib.mp_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc < 0:
            raise NativeError(rc, self.lib.mp_last_error().decode())
        return rc

    def check_name(self, code):
        return self.lib.mp_check_name(code).decode()

    def proof_size(self, m, n):
        return self.lib.mp_proof_size_curve(CURVE_IDS[self.curve], m, n)

    def blake2s(self, data):
        out = (ctypes.c_uint8 * 32)()
        self._chk(self.lib.mp_blake2s(_in(data), len(data), out))
        return bytes(out)

    def setup(self, m, n, seed):
        out = (ctypes.c_uint8 * self.lib.mp_params_size_curve(CURVE_IDS[self.curve], n))()
        self._chk(self.lib.mp_setup(self.h, m, n, _in(seed), out))
        return bytes(out)

    def table(self, m, n, params, shared_key, fb_bits=8):
        return Table(self, m, n, params, shared_key, fb_bits)

    def sync(self):
        self._chk(self.lib.mp_sync(self.h))

    def points_deserialize_dev(self, count, d_data, d_out_wire, d_status):
        """compressed arkworks points in device memory -> wire v1 in device memory; d_status: one int32 per point"""
        self._chk(self.lib.mp_points_deserialize_dev(self.h, count, d_data, d_out_wire, d_status))

    def deck_deserialize_dev(self, decks, cards, d_data, d_out_wire_decks, d_status):
        """`decks` serialised Vec<MaskedCard> of `cards` cards each (device memory) -> wire decks; d_status: one int32 per deck"""
        self._chk(self.lib.mp_deck_deserialize_dev(self.h, decks, cards, d_data, d_out_wire_decks, d_status))

    def profile_enable(self, on=True):
        self._chk(self.lib.mp_profile_enable(self.h, 1 if on else 0))

    def profile_report(self):
        buf = ctypes.create_string_buffer(1 << 16)
        self._chk(self.lib.mp_profile_report(self.h, buf, len(buf)))
        out = {}
        self.last_profile_items = {}      # name -> threads launched (waves / (equation, window) items for the wave kernels)
        for line in buf.value.decode().splitlines():
            f = line.split()
            out[f[0]] = (int(f[1]), float(f[2]))
            if len(f) > 3:
                self.last_profile_items[f[0]] = int(f[3])
        return out


class KeySet:
    """mp_keyset: fixed-base window tables of n aggregate keys of one Table (include/mpshuffle.h, "key sets")"""

    def __init__(self, table, keys):
        self.table, self.lib = table, table.lib
        if len(keys) == 0 or len(keys) % table.pb:
            raise ValueError("keys: expected a whole number of %d-byte points" % table.pb)
        self.size = len(keys) // table.pb
        h = ctypes.c_void_p()
        table.eng._chk(self.lib.mp_keyset_create(table.h, self.size, _in(keys), ctypes.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.mp_keyset_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Table:
    """one mp_table: Parameters + aggregate key with their fixed-base tables in HBM"""

    def __init__(self, eng, m, n, params, shared_key, fb_bits=8):
        self.eng, self.lib, self.m, self.n, self.N = eng, eng.lib, m, n, m * n
        self.fb_bits = fb_bits
        self.params, self.shared_key = bytes(params), (bytes(shared_key) if shared_key is not None else None)
        self.pb = eng.point_bytes
        self.cb = 2 * eng.point_bytes       # bytes of one card (ciphertext)
        if len(self.params) != self.pb * (n + 3) or (self.shared_key is not None and len(self.shared_key) != self.pb):
            raise NativeError(MP_ERR_BAD_ARGUMENT, "parameters / shared key have the wrong length")
        h = ctypes.c_void_p()
        if self.shared_key is None:         # parameters only: a table for keyed batches (one aggregate key per proof)
            eng._chk(self.lib.mp_table_create_params(eng.h, m, n, _in(self.params), fb_bits, ctypes.byref(h)))
        else:
            eng._chk(self.lib.mp_table_create_ex(eng.h, m, n, _in(self.params), _in(self.shared_key), fb_bits, ctypes.byref(h)))
        self.h = h
        self.fb_bits = self.lib.mp_table_window_bits(h)      # (fb_bits = 0: the engine chose by free HBM, as mp_table_create does)
        self.proof_bytes = eng.proof_size(m, n)

    def close(self):
        if getattr(self, "h", None):
            self.lib.mp_table_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- host-buffer API
    def shuffle_and_remask_batch(self, decks, factors, perms, seeds):
        """B proofs; decks: B*N*128 bytes, factors: B*N*32, perms: list of B*N ints, seeds: B*32 -> (decks, proofs, status)"""
        B = len(seeds) // 32
        N = self.N
        self._need("prover seeds", len(seeds), B * 32)
        self._need("decks", len(decks), B * N * self.cb)
        self._need("masking factors", len(factors), B * N * 32)
        self._need("permutations", len(perms), B * N)
        out_d = (ctypes.c_uint8 * (B * N * self.cb))()
        out_p = (ctypes.c_uint8 * (B * self.proof_bytes))()
        st = (ctypes.c_int32 * B)()
        pm = (ctypes.c_uint32 * (B * N))(*perms)
        self.eng._chk(self.lib.mp_shuffle_and_remask_batch(self.h, B, _in(decks), _in(factors), pm, _in(seeds), out_d, out_p, st))
        return bytes(out_d), bytes(out_p), list(st)

    # ---- keyed batches: one aggregate key per proof (keys: B wire points)
    def shuffle_and_remask_batch_keys(self, keys, decks, factors, perms, seeds):
        B = len(seeds) // 32
        N = self.N
        self._need("prover seeds", len(seeds), B * 32)
        self._need("keys", len(keys), B * self.pb)
        self._need("decks", len(decks), B * N * self.cb)
        self._need("masking factors", len(factors), B * N * 32)
        self._need("permutations", len(perms), B * N)
        out_d = (ctypes.c_uint8 * (B * N * self.cb))()
        out_p = (ctypes.c_uint8 * (B * self.proof_bytes))()
        st = (ctypes.c_int32 * B)()
        pm = (ctypes.c_uint32 * (B * N))(*perms)
        self.eng._chk(self.lib.mp_shuffle_and_remask_batch_keys(self.h, B, _in(keys), _in(decks), _in(factors), pm, _in(seeds), out_d, out_p, st))
        return bytes(out_d), bytes(out_p), list(st)

    def verify_shuffle_batch_keys(self, keys, decks, shuffled, proofs):
        N = self.N
        B = len(decks) // (N * self.cb)
        self._need("decks", len(decks), B * N * self.cb)
        self._need("keys", len(keys), B * self.pb)
        self._need("shuffled decks", len(shuffled), len(decks))
        self._need("proofs", len(proofs), B * self.proof_bytes)
        st = (ctypes.c_int32 * B)()
        self.eng._chk(self.lib.mp_verify_shuffle_batch_keys(self.h, B, _in(keys), _in(decks), _in(shuffled), _in(proofs), st))
        return list(st)

    # ---- chain verification: `tables` chains of `links` shuffles; decks: (links + 1) * tables decks (deck j of table t at j * tables + t)
    def verify_shuffle_chain(self, tables, links, decks, proofs, keys=None):
        self._need("decks", len(decks), (links + 1) * tables * self.N * self.cb)
        self._need("proofs", len(proofs), links * tables * self.proof_bytes)
        if keys is not None:
            self._need("keys", len(keys), links * tables * self.pb)
        st = (ctypes.c_int32 * (links * tables))()
        self.eng._chk(self.lib.mp_verify_shuffle_chain(self.h, tables, links, _in(keys) if keys is not None else None, _in(decks),
                                                       _in(proofs), st))
        return list(st)

    def verify_shuffle_chain_dev(self, tables, links, d_keys, d_decks, d_proofs, d_status):
        self.eng._chk(self.lib.mp_verify_shuffle_chain_dev(self.h, tables, links, d_keys, d_decks, d_proofs, d_status))

    # ---- key sets: window tables of many aggregate keys, built once; proofs name their key by index (device arrays of uint32)
    def keyset(self, keys):
        """keys: n wire points back to back (host bytes) -> KeySet (close() it before the table)"""
        return KeySet(self, keys)

    def shuffle_and_remask_batch_keyset_dev(self, ks, B, d_key_index, d_decks, d_factors, d_perms, d_seeds, d_out_decks, d_out_proofs, d_status):
        self.eng._chk(self.lib.mp_shuffle_and_remask_batch_keyset_dev(self.h, ks.h, B, d_key_index, d_decks, d_factors, d_perms, d_seeds,
                                                                      d_out_decks, d_out_proofs, d_status))

    def verify_shuffle_batch_keyset_dev(self, ks, B, d_key_index, d_decks, d_shuffled, d_proofs, d_status):
        self.eng._chk(self.lib.mp_verify_shuffle_batch_keyset_dev(self.h, ks.h, B, d_key_index, d_decks, d_shuffled, d_proofs, d_status))

    def shuffle_and_remask_batch_keys_dev(self, B, d_keys, d_decks, d_factors, d_perms, d_seeds, d_out_decks, d_out_proofs, d_status):
        self.eng._chk(self.lib.mp_shuffle_and_remask_batch_keys_dev(self.h, B, d_keys, d_decks, d_factors, d_perms, d_seeds,
                                                                    d_out_decks, d_out_proofs, d_status))

    def verify_shuffle_batch_keys_dev(self, B, d_keys, d_decks, d_shuffled, d_proofs, d_status):
        self.eng._chk(self.lib.mp_verify_shuffle_batch_keys_dev(self.h, B, d_keys, d_decks, d_shuffled, d_proofs, d_status))

    def verify_shuffle_batch(self, decks, shuffled, proofs):
        N = self.N
        B = len(decks) // (N * self.cb)
        self._need("decks", len(decks), B * N * self.cb)
        self._need("shuffled decks", len(shuffled), len(decks))
        self._need("proofs", len(proofs), B * self.proof_bytes)
        st = (ctypes.c_int32 * B)()
        self.eng._chk(self.lib.mp_verify_shuffle_batch(self.h, B, _in(decks), _in(shuffled), _in(proofs), st))
        return list(st)

    def _need(self, what, got, want):
        if got != want:
            raise NativeError(MP_ERR_BAD_ARGUMENT, "%s: %d bytes/entries given, %d expected" % (what, got, want))

    def shuffle_and_remask(self, deck, factors, perm, seed):
        self._need("deck", len(deck), self.N * self.cb)
        self._need("masking factors", len(factors), self.N * 32)
        self._need("permutation", len(perm), self.N)
        self._need("prover seed", len(seed), 32)
        out_d = (ctypes.c_uint8 * (self.N * self.cb))()
        out_p = (ctypes.c_uint8 * self.proof_bytes)()
        pm = (ctypes.c_uint32 * self.N)(*perm)
        rc = self.lib.mp_shuffle_and_remask(self.h, _in(deck), _in(factors), pm, _in(seed), out_d, out_p)
        self.eng._chk(rc)
        return bytes(out_d), bytes(out_p)

    def verify_shuffle(self, deck, shuffled, proof):
        self._need("deck", len(deck), self.N * self.cb)
        self._need("shuffled deck", len(shuffled), self.N * self.cb)
        self._need("proof", len(proof), self.proof_bytes)
        rc = self.lib.mp_verify_shuffle(self.h, _in(deck), _in(shuffled), _in(proof), len(proof))
        return self.eng._chk(rc)

    def remask_batch(self, cards, factors):
        count = len(cards) // self.cb
        out = (ctypes.c_uint8 * (count * self.cb))()
        self.eng._chk(self.lib.mp_remask_batch(self.h, count, _in(cards), _in(factors), out))
        return bytes(out)

    def msm(self, n_msm, k, scalars, points):
        out = (ctypes.c_uint8 * (n_msm * self.pb))()
        self.eng._chk(self.lib.mp_msm(self.h, n_msm, k, _in(scalars), _in(points), out))
        return bytes(out)

    def commit_batch(self, count, length, values, r):
        out = (ctypes.c_uint8 * (count * self.pb))()
        self.eng._chk(self.lib.mp_commit_batch(self.h, count, length, _in(values), _in(r), out))
        return bytes(out)

    # ---- device-pointer API (ints = HBM addresses, e.g. torch tensor .data_ptr())
    def set_latency_batch(self, B):
        self.eng._chk(self.lib.mp_set_latency_batch(self.h, B))

    def reserve(self, B):
        self.eng._chk(self.lib.mp_reserve(self.h, B))

    def shuffle_and_remask_batch_dev(self, B, d_decks, d_factors, d_perms, d_seeds, d_out_decks, d_out_proofs, d_status):
        self.eng._chk(self.lib.mp_shuffle_and_remask_batch_dev(self.h, B, d_decks, d_factors, d_perms, d_seeds, d_out_decks, d_out_proofs, d_status))

    def verify_shuffle_batch_dev(self, B, d_decks, d_shuffled, d_proofs, d_status):
        self.eng._chk(self.lib.mp_verify_shuffle_batch_dev(self.h, B, d_decks, d_shuffled, d_proofs, d_status))

    def sigma_prove_batch(self, nbases, bases, publics, witness, fs_init, seeds):
        B = len(witness) // 32
        out = (ctypes.c_uint8 * (B * (nbases * self.pb + 32)))()
        st = (ctypes.c_int32 * B)()
        self.eng._chk(self.lib.mp_sigma_prove_batch(self.h, B, nbases, _in(bases), _in(publics), _in(witness), _in(fs_init),
                                                    _in(seeds), out, st))
        return bytes(out), list(st)

    def sigma_verify_batch(self, nbases, bases, publics, proofs, fs_init):
        B = len(proofs) // (nbases * self.pb + 32)
        st = (ctypes.c_int32 * B)()
        self.eng._chk(self.lib.mp_sigma_verify_batch(self.h, B, nbases, _in(bases), _in(publics), _in(proofs), _in(fs_init), st))
        return list(st)

    def set_io_chunk(self, proofs):
        """proofs per pipelined chunk of the host-buffer entry points (0 = default 65536)"""
        self.eng._chk(self.lib.mp_set_io_chunk(self.h, proofs))

    def set_merged_verify(self, on=True):
        """verification strategy: merged screening pass first (default) or always equation by equation"""
        self.eng._chk(self.lib.mp_set_merged_verify(self.h, 1 if on else 0))

    def set_bucket_min(self, terms):
        """variable-base MSMs of at least `terms` terms run on the bucket-method kernel (default 2048; 0 = never)"""
        self.eng._chk(self.lib.mp_set_bucket_min(self.h, terms))

    def set_bucket_bits(self, bits):
        """window width of the bucket method: 8 .. 13, or 0 = by the size of the MSM (default: 11 from 40 000 terms on, 12 from 100 000, 13 from 200 000)"""
        self.eng._chk(self.lib.mp_set_bucket_bits(self.h, bits))

    VALIDATED_DECKS, VALIDATED_SHUFFLED, VALIDATED_PROOFS = 1, 2, 4

    def set_validated(self, what):
        """inputs the caller has validated once already (OR of VALIDATED_*): their subgroup test is not repeated in every call"""
        self.eng._chk(self.lib.mp_set_validated(self.h, what))

    def deck_validate_dev(self, decks, d_wire_decks, d_status):
        """wire-v1 decks in device memory -> one int32 per deck (0 / MP_ERR_BAD_ENCODING): range, curve equation, prime-order subgroup"""
        self.eng._chk(self.lib.mp_deck_validate_dev(self.h, decks, d_wire_decks, d_status))

    def set_bucket_split(self, min_bits):
        """windows of at least `min_bits` bits run sort / additions / reduction as three kernels (default 12; 10 .. 15 = never)"""
        self.eng._chk(self.lib.mp_set_bucket_split(self.h, min_bits))

    def set_chain_max_links(self, links):
        """chain verification: at most `links` links per chain equation (0 = as many as fit)"""
        self.eng._chk(self.lib.mp_set_chain_max_links(self.h, links))

    def set_chain_slice(self, tables_per_pass):
        """chain verification in passes of this many tables (0 = one pass if its workspace fits the free memory)"""
        self.eng._chk(self.lib.mp_set_chain_slice(self.h, tables_per_pass))

    def chain_group_size(self, tables, links, keyed=False):
        """tables per chain equation a pass of `tables` tables x `links` links takes under the current settings"""
        return int(self.lib.mp_chain_group_size(self.h, tables, links, 1 if keyed else 0))

    def chain_last_slice(self):
        """tables per pass of the last chain verification call on this table"""
        return int(self.lib.mp_chain_last_slice(self.h))

    def set_chain_group(self, tables_per_equation):
        """chain verification: tables per chain equation (0 = by size, 1 = every table on its own)"""
        self.eng._chk(self.lib.mp_set_chain_group(self.h, tables_per_equation))

    def set_transcript_lanes(self, lanes):
        """lanes per Fiat-Shamir transcript hash: 1, 4, or 0 = by batch size (4 up to 32 768 proofs)"""
        self.eng._chk(self.lib.mp_set_transcript_lanes(self.h, lanes))

    def set_group_lanes(self, lanes):
        """lanes per group operation of the MSM dependency chains: 1, 4, or 0 = by batch size (4 for a few dozen proofs)"""
        self.eng._chk(self.lib.mp_set_group_lanes(self.h, lanes))

    def set_work_split(self, split):
        """every batch takes work split `split` (0 throughput, 1 latency, 2 medium, 3 finest, 4 wide, 5 small); -1 = by batch size"""
        self.eng._chk(self.lib.mp_set_work_split(self.h, split))

    def group_size(self, B):
        """proofs per group of the screening pass for a batch of B (0: per-proof screen)"""
        return self.lib.mp_group_size(self.h, B)

    def set_group_verify(self, points_per_group=243712, min_batch=6144):
        """screening pass of large batches: one equation of ~points_per_group points per group of proofs on the bucket kernels (a proof
        brings 4N + 11m + 8 points; 0 = off; up to 65 535: at most that many and one wave per window, as in rounds 4-5; the default:
        equations of up to 1 024 52-card proofs on the split pipeline for batches of 32 768 proofs and more)"""
        self.eng._chk(self.lib.mp_set_group_verify(self.h, points_per_group, min_batch))

    def set_group_refine(self, points_per_subgroup=0, min_subgroups=0):
        """what a failing group costs: its members go through equations of sub-groups of ~points_per_subgroup points (0 = an eighth of
        the group equation's) when there are at least min_subgroups of them (0 = 128), else straight to the per-equation pass"""
        self.eng._chk(self.lib.mp_set_group_refine(self.h, points_per_subgroup, min_subgroups))

    def set_group_adapt(self, on=True):
        """groups that shrink under sustained rejection (default on): more than a fifth of a call's groups failing halves the next call's
        group size, fewer than 4 % restore it step by step; False pins the default size"""
        self.eng._chk(self.lib.mp_set_group_adapt(self.h, 1 if on else 0))

    def reverified_count(self):
        """proofs that have taken a per-equation pass on this table because a screen could not clear them"""
        return int(self.lib.mp_reverified_count(self.h))

    def set_pipeline(self, depth=1):
        """depth >= 1: device-resident verify calls run on the context's second lane beside the next prove call and do not wait for
        their screening verdict (include/mpshuffle.h: mp_set_pipeline); their inputs must stay untouched until `depth` further verify
        calls or a sync have returned; 0 = off"""
        self.eng._chk(self.lib.mp_set_pipeline(self.h, int(depth)))

    def set_plan_params(self, split, fixed_terms, var_terms, table_group, norm_chunk, window_lanes=1):
        """sizes of work split `split`: terms per fixed-base / variable-base sub-job, bases per table lane, points per inversion,
        lanes per variable-base sub-job (window split)"""
        self.eng._chk(self.lib.mp_set_plan_params(self.h, split, fixed_terms, var_terms, table_group, norm_chunk, window_lanes))

    def set_plan_thresholds(self, finest, small, latency, medium, wide):
        """largest batch that takes the finest / small / latency / medium / wide split"""
        self.eng._chk(self.lib.mp_set_plan_thresholds(self.h, finest, small, latency, medium, wide))

    def set_toom_cook(self, on=True):
        """3 <= m <= 8: Toom-Cook (default) or Karatsuba evaluation of the multi-exponentiation diagonals"""
        self.eng._chk(self.lib.mp_set_toom_cook(self.h, 1 if on else 0))

    def set_subgroup_check(self, on=True):
        """curves with a cofactor: test every wire point for membership in the prime-order subgroup (default on)"""
        self.eng._chk(self.lib.mp_set_subgroup_check(self.h, 1 if on else 0))

    def plan_stats(self):
        v = (ctypes.c_uint64 * 16)()
        self.eng._chk(self.lib.mp_plan_stats(self.h, v))
        keys = ["fixed_terms", "var_terms", "fixed_jobs", "var_jobs", "table_bases", "combine_terms"]
        out = {"prove": dict(zip(keys, v[0:6])), "verify": dict(zip(keys, v[6:12]))}
        out.update(var_windows=v[12], fixed_windows=v[13], N=v[14] & 0xFFFFFFFF, toom_points_m=v[14] >> 32)
        # MSMs on the bucket-method kernel (prove + verify together): terms and jobs; 8-bit windows
        out.update(bucket_terms=v[15] & 0xFFFFFFFF, bucket_jobs=v[15] >> 32)
        return out

    def work_census(self):
        v = [ctypes.c_uint64() for _ in range(4)]
        self.eng._chk(self.lib.mp_work_census(self.h, *[ctypes.byref(x) for x in v]))
        return dict(prove_terms=v[0].value, verify_terms=v[1].value, prove_point_ops=v[2].value, verify_point_ops=v[3].value)
