"""Condense `make -C tools_amd/csrc resource-usage` (clang's -Rpass-analysis=kernel-resource-usage remarks on stderr) into one line per kernel.

    make -C tools_amd/csrc resource-usage 2> /tmp/ru.txt >/dev/null; python tools/resource_usage_table.py /tmp/ru.txt > profiles/rNN_resource_usage.txt
"""
import re, subprocess, sys

KEYS = ["TotalSGPRs", "VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "SGPRs Spill", "VGPRs Spill", "LDS Size [bytes/block]"]


def main(path):
    rows, cur = [], None
    for line in open(path):
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\S+) \[-Rpass", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = m.group(2)
    names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
    print("# one line per kernel of libpsf_mi355x.so (gfx950), from clang's kernel-resource-usage remarks; tools/resource_usage_table.py")
    print("# sgpr vgpr agpr scratch_B/lane occupancy sgpr_spill vgpr_spill lds_B  kernel")
    bad = []
    for r, n in zip(rows, names):
        n = re.sub(r"\(.*$", "", n).replace("void ", "")
        v = [r.get(k, "?") for k in KEYS]
        print("%4s %4s %4s %6s %3s %5s %5s %7s  %s" % (*v, n))
        if v[3] not in ("0", "?") or v[5] not in ("0", "?") or v[6] not in ("0", "?"):
            bad.append((n, v[3], v[5], v[6]))
    print("#\n# kernels with scratch or spills (scratch_B/lane, sgpr_spill, vgpr_spill):")
    for n, s, a, b in bad:
        print("#   %-70s %6s %5s %5s" % (n, s, a, b))
    print("# %d kernels, %d with scratch or spills" % (len(rows), len(bad)))


if __name__ == "__main__":
    main(sys.argv[1])
