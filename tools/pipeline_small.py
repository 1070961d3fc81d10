"""Throughput of a LOOP of small samp_p calls at C3 (device pointers, no synchronisation between calls): plain against PSF_PIPELINE=1 (experiments build: the normals
and the product of call i + 1 on one stream, the stages behind the product of call i on another).  At 4096 preimages the overlap is zero-sum (the FP64 matrix product
holds the vector pipe); a product of one to 64 preimages is bound by reading the factor and leaves the vector pipe idle.   python tools/pipeline_small.py [calls=300]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def child(mode, calls):
    import torch, bench
    import tools_amd as T
    _, n, q, r, s, _ = bench.CONFIGS["c3"]
    gp = T.GadgetParameters.init_default(n, q)
    psf = T.PSFPerturbation(gp, r, s)
    psf.trap_gen(1)
    m = gp.m_bar + gp.n * gp.k
    dev = torch.device("cuda:0")
    u = (torch.randint(0, 2**62, (64, n), dtype=torch.int64, generator=torch.Generator().manual_seed(5)) % q).to(dev)
    stream = torch.cuda.current_stream().cuda_stream
    out = {"mode": mode}
    for B in (1, 16, 64):
        es = [torch.zeros((B, m), dtype=torch.int64, device=dev) for _ in range(2)]
        for i in range(4): psf.samp_p_dev(u.data_ptr(), es[i & 1].data_ptr(), B, seed=9, first_index=1000 + i * B, stream=stream)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for i in range(calls): psf.samp_p_dev(u.data_ptr(), es[i & 1].data_ptr(), B, seed=9, first_index=5000 + i * B, stream=stream)
        ev1.record(); torch.cuda.synchronize()
        assert psf.last_status() == 0
        out[f"ms_per_call_b{B}"] = round(ev0.elapsed_time(ev1) / calls, 4)
        out[f"checksum_b{B}"] = int(es[(calls - 1) & 1].sum().item())
    print(json.dumps(out), flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]))
    else:
        calls = int(sys.argv[1]) if len(sys.argv) > 1 else 300
        for mode in ("0", "1", "0", "1"):
            env = dict(os.environ, PSF_LIB=os.path.join(ROOT, "tools_amd", "lib", "libpsf_mi355x_exp.so"), PSF_PIPELINE=mode)
            subprocess.run([sys.executable, __file__, "--child", mode, str(calls)], env=env, check=True)
