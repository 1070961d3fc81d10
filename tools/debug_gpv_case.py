"""debug: a random GPV configuration where the GPU reported PSF_ERR_SAMPLER and the oracle did not"""
import ctypes as C, sys, os, math
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools_amd as T
from tools_amd._ffi import lib
from tools_amd.psf import _p
from oracle import oracle

PRIMES = [257, 3329, 7681, 12289, 65537, 1073741789, 2**31 - 1, 2**61 - 1]
def draw_modulus(rng):
    kind = rng.integers(0, 3)
    if kind == 0: return int(2 ** rng.integers(4, 61))
    if kind == 1: return int(PRIMES[rng.integers(0, len(PRIMES))])
    return int(rng.integers(17, 2**20)) | 1
case = int(sys.argv[1]) if len(sys.argv) > 1 else 15
rng = np.random.default_rng(2000 + case)
q = draw_modulus(rng)
n = int(rng.integers(2, 40 if q < 2**24 else 12))
s = float(rng.choice([8.0, 30.0, 240.0, 1000.0])) * (1.0 if q < 2**30 else 4.0)
B = int(rng.choice([1, 3, 4, 5, 8, 9, 64, 130]))
print("case", case, "n", n, "q", q, "s", s, "B", B)
psf = T.PSFGPV(T.GadgetParameters.init_default(n, q), s)
A, (bt, gt) = psf.trap_gen(200 + case)
orc = oracle.PSFGPV(oracle.gadget_params_default(n, q), s)
assert orc.load_key(A, bt, gt) == 0
print("two_pass", psf.two_pass, orc.two_pass, "m", psf.m)
u = oracle.uniform_targets(case, B, n, q)
first = int(rng.integers(0, 2**40))
e_ref = orc.samp_p(9 + case, u, first_index=first)
for b in range(B):
    e = np.zeros((1, psf.m), dtype=np.int64)
    ub = np.ascontiguousarray(u[b:b+1].astype(np.uint64))
    rc = lib().psfgpv_samp_p(psf._h, C.c_uint64(9 + case), C.c_uint64(first + b), C.c_size_t(1), _p(ub, C.c_uint64), _p(e, C.c_int64))
    same = (e[0] == e_ref[b]).all()
    print("row", b, "rc", rc, "equal to oracle:", same, "" if same else ("first diff at %d: gpu %d oracle %d" % (int(np.nonzero(e[0] != e_ref[b])[0][0]), e[0][np.nonzero(e[0] != e_ref[b])[0][0]], e_ref[b][np.nonzero(e[0] != e_ref[b])[0][0]])))
    if rc != 0:
        st = (C.c_uint64 * 2)()
        tr = orc.samp_p_trace(9 + case, u[b], index=first + b)
        cen = np.asarray(tr[2]); z = np.asarray(tr[3])
        nr = np.sqrt((gt.astype(np.float64) ** 2).sum(axis=1))
        sp = s / nr
        frac = np.abs(cen - np.round(cen))
        worst = np.argsort(-(frac / sp))[:5]
        print("   oracle trace: |z| max", np.abs(z).max(), "rows with the hardest draws (i, s', centre, dist to nearest integer):", [(int(i), float(sp[i]), float(cen[i]), float(frac[i])) for i in worst])
for force in (0, 1):
    os.environ["PSF_NP_TWO_PASS"] = str(force)
    p2 = T.PSFGPV(T.GadgetParameters.init_default(n, q), s)
    p2.load_key(A, bt, gt)
    o2 = oracle.PSFGPV(oracle.gadget_params_default(n, q), s); o2.set_two_pass(force); o2.load_key(A, bt, gt)
    e = np.zeros((B, psf.m), dtype=np.int64)
    uu = np.ascontiguousarray(u.astype(np.uint64))
    rc = lib().psfgpv_samp_p(p2._h, C.c_uint64(9 + case), C.c_uint64(first), C.c_size_t(B), _p(uu, C.c_uint64), _p(e, C.c_int64))
    try:
        er = o2.samp_p(9 + case, u, first_index=first); orc_rc = 0
    except RuntimeError as ex:
        er = None; orc_rc = str(ex)
    print("forced two_pass =", force, ": gpu rc", rc, "oracle", orc_rc, "equal", None if er is None else bool((e == er).all()))
