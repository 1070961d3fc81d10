"""A/B of one stage of a small-batch call at C3 under experiment switches (experiments build): HIP-event time of the stage's launches (the handle's own timers), the
median call, and the rows of every variant compared bit for bit with the first.
   python tools/tail_ab.py [--config=bench64] <timer name, e.g. k_recombine> <sizes, comma separated> <name>:<K=V,K=V> [<name>:<K=V> ...]      ("name:" alone = no switch)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("PSF_LIB", os.path.join(ROOT, "tools_amd", "lib", "libpsf_mi355x_exp.so"))
import torch  # noqa: E402
import bench  # noqa: E402
import tools_amd as T  # noqa: E402

def main():
    cfg = "c3"
    if len(sys.argv) > 1 and sys.argv[1].startswith("--config="):      # another PSFPerturbation configuration of bench.CONFIGS (bench64, c1, c3prime)
        cfg = sys.argv.pop(1).split("=", 1)[1]
    key = sys.argv[1]
    sizes = [int(x) for x in sys.argv[2].split(",")]
    variants = []
    for a in sys.argv[3:]:
        name, _, kv = a.partition(":")
        variants.append((name, dict(x.split("=", 1) for x in kv.split(",") if x)))
    allkeys = sorted({k for _, env in variants for k in env})
    _, n, q, r, s, _ = bench.CONFIGS[cfg]
    gp = T.GadgetParameters.init_default(n, q)
    psf = T.PSFPerturbation(gp, r, s)
    psf.trap_gen(1)
    m = gp.m_bar + gp.n * gp.k
    dev = torch.device("cuda:0")
    u = (torch.randint(0, 2**62, (max(sizes), n), dtype=torch.int64) % q).to(dev)
    stream = torch.cuda.current_stream().cuda_stream
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for B in sizes:
        ref = None
        row = {"B": B}
        for name, env in variants:
            for k in allkeys:
                os.environ.pop(k, None)
            os.environ.update(env)
            e = torch.zeros((B, m), dtype=torch.int64, device=dev)
            call = lambda: psf.samp_p_dev(u.data_ptr(), e.data_ptr(), B, seed=9, first_index=1000, stream=stream)
            call(); call()
            ts, cs = [], []
            for _ in range(9):
                torch.cuda.synchronize(); ev0.record(); call(); ev1.record(); torch.cuda.synchronize()
                cs.append(ev0.elapsed_time(ev1))
                psf.enable_timing(True); call(); tm = dict(psf.get_timing()); psf.enable_timing(False)
                ts.append(tm.get(key, 0.0))
            ts.sort(); cs.sort()
            row[name] = [round(ts[len(ts) // 2], 4), round(cs[len(cs) // 2], 4)]
            assert psf.last_status() == 0
            if ref is None: ref = e.clone()
            else: assert (ref == e).all(), (B, name)
        print(json.dumps(row), flush=True)

if __name__ == "__main__":
    main()
