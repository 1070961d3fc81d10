"""The gadget walk at small and mid-size batches (C3): the default choice of the batch size, then every form forced -- one wave per problem (k_gadget_wave), sixteen
lanes (k_gadget_quad<., 16>), four lanes (k_gadget_quad<., 4>), round 4's sixteen-lane kernel with exact attempts (k_gadget_wave16), the queue kernel with 32 problems per
wave (PSF_GQ_P) and with its own choice; experiments build; rows compared bit for bit.
   python tools/gadget_mid_ab.py [sizes ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("PSF_LIB", os.path.join(ROOT, "tools_amd", "lib", "libpsf_mi355x_exp.so"))
import torch  # noqa: E402
import bench  # noqa: E402
import tools_amd as T  # noqa: E402

def main():
    sizes = [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024]
    _, n, q, r, s, _ = bench.CONFIGS["c3"]
    gp = T.GadgetParameters.init_default(n, q)
    psf = T.PSFPerturbation(gp, r, s)
    psf.trap_gen(1)
    m = gp.m_bar + gp.n * gp.k
    dev = torch.device("cuda:0")
    u = (torch.randint(0, 2**62, (max(sizes), n), dtype=torch.int64) % q).to(dev)
    stream = torch.cuda.current_stream().cuda_stream
    for B in sizes:
        ref = None
        row = {"B": B}
        OFF = {"PSF_GADGET_WAVE": "0", "PSF_GADGET_ROW": "0", "PSF_GADGET_QUAD": "0", "PSF_GADGET_WAVE16": "0"}
        for name, env in [("default", {}), ("wave", dict(OFF, PSF_GADGET_WAVE="100000000")), ("row", dict(OFF, PSF_GADGET_ROW="100000000")), ("quad", dict(OFF, PSF_GADGET_QUAD="100000000")),
                          ("wave16", dict(OFF, PSF_GADGET_WAVE16="100000000")), ("q32", dict(OFF, PSF_GADGET_QP="32", PSF_GQ_P="32")), ("qauto", dict(OFF))]:
            for k in ("PSF_GADGET_WAVE16", "PSF_GQ_P", "PSF_GADGET_QUAD", "PSF_GADGET_WAVE", "PSF_GADGET_ROW", "PSF_GADGET_QP"):
                os.environ.pop(k, None)
            os.environ.update(env)
            e = torch.zeros((B, m), dtype=torch.int64, device=dev)
            call = lambda: psf.samp_p_dev(u.data_ptr(), e.data_ptr(), B, seed=9, first_index=1000, stream=stream)
            call(); call()
            ts = []
            for _ in range(7):
                psf.enable_timing(True); call(); tm = dict(psf.get_timing()); psf.enable_timing(False)
                ts.append(tm["k_gadget"])
            ts.sort()
            row[name] = round(ts[len(ts) // 2], 4)
            if ref is None: ref = e.clone()
            else: assert (ref == e).all(), (B, name)
        print(json.dumps(row), flush=True)

if __name__ == "__main__":
    main()
