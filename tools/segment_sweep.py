"""Positions per wave of the normals kernel (PSF_NR_SEG) and samples per wave of the rounding kernel (PSF_PRL_SEG) at small batches of C3: HIP-event times of the two
kernels; experiments build; rows compared bit for bit.    python tools/segment_sweep.py [sizes ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("PSF_LIB", os.path.join(ROOT, "tools_amd", "lib", "libpsf_mi355x_exp.so"))
import torch  # noqa: E402
import bench  # noqa: E402
import tools_amd as T  # noqa: E402

def main():
    sizes = [int(x) for x in sys.argv[1:]] or [4, 16, 64, 256]
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
        for var, key in (("PSF_NR_SEG", "k_normals"), ("PSF_PRL_SEG", "k_perturb_round")):
            row = {"B": B, "switch": var}
            for seg in (0, 64, 128, 256, 512, 1024, 2048, 4096):
                os.environ.pop("PSF_NR_SEG", None); os.environ.pop("PSF_PRL_SEG", None)
                if seg: os.environ[var] = str(seg)
                e = torch.zeros((B, m), dtype=torch.int64, device=dev)
                call = lambda: psf.samp_p_dev(u.data_ptr(), e.data_ptr(), B, seed=9, first_index=1000, stream=stream)
                call(); call()
                ts = []
                for _ in range(7):
                    psf.enable_timing(True); call(); tm = dict(psf.get_timing()); psf.enable_timing(False)
                    ts.append(tm.get(key, 0.0))
                ts.sort()
                row["default" if seg == 0 else str(seg)] = round(ts[len(ts) // 2], 4)
                if ref is None: ref = e.clone()
                else: assert (ref == e).all(), (B, var, seg)
            print(json.dumps(row), flush=True)

if __name__ == "__main__":
    main()
