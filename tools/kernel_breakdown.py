"""Per-stage HIP-event times (the handle's own timers, median of five calls) of one samp_p_dev call at C3 for a list of batch sizes: where a batch size's time goes.
   python tools/kernel_breakdown.py [--config=bench64] 64 128 704 768 ..."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
import tools_amd as T
cfg = "c3"
if len(sys.argv) > 1 and sys.argv[1].startswith("--config="): cfg = sys.argv.pop(1).split("=", 1)[1]
_, n, q, r, s, _ = bench.CONFIGS[cfg]
gp = T.GadgetParameters.init_default(n, q); psf = T.PSFPerturbation(gp, r, s); psf.trap_gen(1)
m = gp.m_bar + gp.n * gp.k; dev = torch.device("cuda:0")
sizes=[int(x) for x in sys.argv[1:]]
u = (torch.randint(0, 2**62, (max(sizes), n), dtype=torch.int64) % q).to(dev)
st = torch.cuda.current_stream().cuda_stream
for B in sizes:
    e = torch.zeros((B, m), dtype=torch.int64, device=dev)
    call = lambda: psf.samp_p_dev(u.data_ptr(), e.data_ptr(), B, seed=9, first_index=1000, stream=st)
    call(); call()
    acc={}
    for _ in range(5):
        psf.enable_timing(True); call(); tm=dict(psf.get_timing()); psf.enable_timing(False)
        for k,v in tm.items(): acc.setdefault(k,[]).append(v)
    print(B, {k: round(sorted(v)[2],3) for k,v in acc.items()}, flush=True)
