"""A/B of the small-batch product at C3: one-wave tasks (k_trmm_stream, PSF_STREAM_WG=0) against 64 x 64 tiles with LDS-shared operands (k_trmm_stream_wg).
Experiments build; one key, the switch is read per call.  For every batch size: the rows of the two forms compared bit for bit, the median call and the product's
own HIP-event time.    python tools/stream_wg_ab.py [config=c3] [reps=15] [sizes ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("PSF_LIB", os.path.join(ROOT, "tools_amd", "lib", "libpsf_mi355x_exp.so"))
import torch  # noqa: E402
import bench  # noqa: E402
import tools_amd as T  # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 15
    sizes = [int(x) for x in sys.argv[3:]] or [17, 32, 33, 48, 64, 65, 100, 128, 192, 256, 512, 1024]
    _, n, q, r, s, _ = bench.CONFIGS[cfg]
    gp = T.GadgetParameters.init_default(n, q)
    psf = T.PSFPerturbation(gp, r, s)
    psf.trap_gen(1)
    m = gp.m_bar + gp.n * gp.k
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(5)
    u = (torch.randint(0, 2**62, (max(sizes), n), generator=g, dtype=torch.int64) % q).to(dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    stream = torch.cuda.current_stream().cuda_stream
    rows = []
    for B in sizes:
        res = {}
        for form in ("wave", "wg"):
            if form == "wave":
                os.environ["PSF_STREAM_WG"] = "0"; os.environ["PSF_STREAM_WG32"] = "0"
            else:
                os.environ.pop("PSF_STREAM_WG32", None); os.environ["PSF_STREAM_WG"] = "33"; os.environ["PSF_STREAM_WG_MAX"] = "1024"
            e = torch.zeros((B, m), dtype=torch.int64, device=dev)
            call = lambda: psf.samp_p_dev(u.data_ptr(), e.data_ptr(), B, seed=9, first_index=1000, stream=stream)
            call(); call()
            ts = []
            for _ in range(reps):
                torch.cuda.synchronize()
                ev0.record(); call(); ev1.record()
                torch.cuda.synchronize()
                ts.append(ev0.elapsed_time(ev1))
            ts.sort()
            psf.enable_timing(True)
            call()
            tm = dict(psf.get_timing())
            psf.enable_timing(False)
            assert psf.last_status() == 0
            res[form] = (ts[len(ts) // 2], ts[0], tm.get("k_trmm_f64"), e.clone())
        same = bool((res["wave"][3] == res["wg"][3]).all())
        row = {"B": B, "same_bits": same, "wave_call_ms": round(res["wave"][0], 4), "wg_call_ms": round(res["wg"][0], 4),
               "wave_product_ms": round(res["wave"][2], 4), "wg_product_ms": round(res["wg"][2], 4)}
        rows.append(row)
        print(json.dumps(row), flush=True)
    os.environ.pop("PSF_STREAM_WG", None)
    assert all(r["same_bits"] for r in rows), "the two forms differ"


if __name__ == "__main__":
    main()
