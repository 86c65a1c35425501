"""ctypes binding of oracle/libmporacle.so (the C++ CPU restatement).  TEST INFRASTRUCTURE ONLY:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libmporacle.so")
CURVE_IDS = {"stark": 0, "bn254": 1, "secp256k1": 2, "bls12_377": 3}
CHECK_NAMES = {0: "Ok", 1: "Hadamard Product (5.1)", 2: "Zero Argument (5.2)",
               3: "Single Value Product (5.3)", 4: "Multi-Exponentiation Argument (4)"}
CHECK_NAMES_ALL = {**CHECK_NAMES, 5: "Schnorr Identification", 6: "Chaum-Pedersen"}


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _load():
    if not os.path.exists(_LIB):
        build()
    lib = ctypes.CDLL(_LIB)
    lib.mpo_proof_size.restype = ctypes.c_size_t
    lib.mpo_proof_size.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
    lib.mpo_proof_size_curve.restype = ctypes.c_size_t
    lib.mpo_proof_size_curve.argtypes = [ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32]
    lib.mpo_point_size.restype = ctypes.c_size_t
    lib.mpo_point_size.argtypes = [ctypes.c_int]
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def _buf(n):
    return (ctypes.c_uint8 * n)()


def _in(b):
    return (ctypes.c_uint8 * len(b)).from_buffer_copy(bytes(b)) if len(b) else (ctypes.c_uint8 * 1)()


def proof_size(m, n, curve="stark"):
    return lib().mpo_proof_size_curve(CURVE_IDS[curve], m, n)


def point_size(curve):
    """wire bytes of one affine point: 64, or 96 on BLS12-377"""
    return lib().mpo_point_size(CURVE_IDS[curve])


def gen_inputs(curve, m, n, seed):
    """-> dict(params, pk, deck, rho, perm, prover_seed) in boundary (wire) bytes"""
    N = m * n
    pw = point_size(curve)
    params, pk, deck, rho, ps = _buf(pw * (n + 3)), _buf(pw), _buf(2 * pw * N), _buf(32 * N), _buf(32)
    perm = (ctypes.c_uint32 * N)()
    rc = lib().mpo_gen_inputs(CURVE_IDS[curve], m, n, ctypes.c_uint64(seed), params, pk, deck, rho, perm, ps)
    assert rc == 0, rc
    return dict(params=bytes(params), pk=bytes(pk), deck=bytes(deck), rho=bytes(rho), perm=list(perm),
                prover_seed=bytes(ps))


def shuffle_and_remask(curve, m, n, params, pk, deck, rho, perm, prover_seed):
    N = m * n
    out_deck, out_proof = _buf(2 * point_size(curve) * N), _buf(proof_size(m, n, curve))
    p = (ctypes.c_uint32 * N)(*perm)
    rc = lib().mpo_shuffle_and_remask(CURVE_IDS[curve], m, n, _in(params), _in(pk), _in(deck), _in(rho), p,
                                      _in(prover_seed), out_deck, out_proof)
    if rc != 0:
        raise ValueError("oracle shuffle_and_remask failed: %d" % rc)
    return bytes(out_deck), bytes(out_proof)


def verify_shuffle(curve, m, n, params, pk, deck, shuffled, proof):
    assert len(proof) == proof_size(m, n, curve)
    return lib().mpo_verify_shuffle(CURVE_IDS[curve], m, n, _in(params), _in(pk), _in(deck), _in(shuffled), _in(proof))


def remask_deck(curve, G, pk, deck, rho, perm=None):
    N = len(deck) // (2 * point_size(curve))
    out = _buf(2 * point_size(curve) * N)
    p = (ctypes.c_uint32 * N)(*perm) if perm is not None else None
    rc = lib().mpo_remask_deck(CURVE_IDS[curve], _in(G), _in(pk), _in(deck), ctypes.c_size_t(N), _in(rho), p, out)
    assert rc == 0, rc
    return bytes(out)


def msm(curve, scalars, points, algo=0):
    n = len(scalars) // 32
    out = _buf(point_size(curve))
    rc = lib().mpo_msm(CURVE_IDS[curve], _in(scalars), _in(points), ctypes.c_size_t(n), algo, out)
    assert rc == 0, rc
    return bytes(out)


def commit(curve, n, params, v, r):
    out = _buf(point_size(curve))
    rc = lib().mpo_commit(CURVE_IDS[curve], n, _in(params), _in(v), ctypes.c_size_t(len(v) // 32), _in(r), out)
    assert rc == 0, rc
    return bytes(out)


def fs_challenges(curve, init, absorb, count):
    out = _buf(32 * count)
    a = _in(absorb) if absorb is not None else None
    rc = lib().mpo_fs_challenges(CURVE_IDS[curve], _in(init), ctypes.c_size_t(len(init)), a,
                                 ctypes.c_size_t(len(absorb) if absorb is not None else 0), ctypes.c_size_t(count), out)
    assert rc == 0
    return [int.from_bytes(bytes(out[32 * i:32 * i + 32]), "little") for i in range(count)]


def on_curve(curve, pt):
    return lib().mpo_on_curve(CURVE_IDS[curve], _in(pt))


def sigma_prove(curve, nbases, bases, publics, x, fs_init, seed):
    """Schnorr (1 base) / Chaum-Pedersen (2 bases) proof; fs_init = the bytes the FiatShamirRng is seeded from"""
    out = _buf(point_size(curve) * nbases + 32)
    rc = lib().mpo_sigma_prove(CURVE_IDS[curve], nbases, _in(bases), _in(publics), _in(x), _in(fs_init),
                               ctypes.c_size_t(len(fs_init)), _in(seed), out)
    assert rc == 0, rc
    return bytes(out)


def sigma_verify(curve, nbases, bases, publics, proof, fs_init):
    return lib().mpo_sigma_verify(CURVE_IDS[curve], nbases, _in(bases), _in(publics), _in(proof), _in(fs_init),
                                  ctypes.c_size_t(len(fs_init)))


def blake2s(data):
    out = _buf(32)
    lib().mpo_blake2s(_in(data), ctypes.c_size_t(len(data)), out)
    return bytes(out)


def chacha20_block(key, counter):
    out = (ctypes.c_uint32 * 16)()
    lib().mpo_chacha20_block(_in(key), ctypes.c_uint64(counter), out)
    return list(out)


def bench(curve, m, n, seed, iters):
    """-> (prove_seconds, verify_seconds) summed over `iters` prove+verify pairs, single thread"""
    p, v = ctypes.c_double(), ctypes.c_double()
    rc = lib().mpo_bench(CURVE_IDS[curve], m, n, ctypes.c_uint64(seed), iters, ctypes.byref(p), ctypes.byref(v))
    assert rc == 0, rc
    return p.value, v.value
