"""Fold the counter_collection CSVs of two separate rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of the same bench
command into per-kernel HBM bytes per launch, with the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE counts
128-B read requests as 64 B: double it; WRITE_SIZE is exact).  Kernels are keyed by their FULL name up to the parameter
list - template arguments included - so that the instantiations of one kernel template stay apart (round 3 keyed by the
first 60 characters, which folded gemm_nt_seam_kernel<0|1|3, ...> into one entry).
usage: pmc_traffic.py <dir with the CSVs> <out.json>"""
import csv, glob, json, os, sys
from collections import defaultdict


def kernel_key(name):
    """`void (anonymous namespace)::(anonymous namespace)::gemm_nt_seam_kernel<1, 8, 0, true>((anonymous namespace)::GemmNtArgs)`
    -> `gemm_nt_seam_kernel<1, 8, 0, true>`: return type, anonymous namespaces and the parameter list dropped, template arguments kept.
    Mangled names (no demangler in the loop) are kept whole."""
    n = name.strip()
    if n.startswith("_Z"):
        return n
    if n.startswith("void "):
        n = n[5:]
    n = n.replace("(anonymous namespace)::", "")
    depth = 0
    for i, ch in enumerate(n):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return n[:i].strip()
    return n


def fold(src):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for path in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                name = row.get("Kernel_Name") or row.get("Kernel Name") or ""
                cname = row.get("Counter_Name") or row.get("Counter Name") or ""
                val = float(row.get("Counter_Value") or row.get("Counter Value") or 0)
                if cname in ("FETCH_SIZE", "WRITE_SIZE"):
                    a = acc[kernel_key(name)][cname]; a[0] += val; a[1] += 1
    res = {}
    for k, d in acc.items():
        f, w = d.get("FETCH_SIZE", [0, 0]), d.get("WRITE_SIZE", [0, 0])
        if not f[1] or not w[1]:
            continue
        fk, wk = f[0] / f[1], w[0] / w[1]
        res[k] = {"launches": int(max(f[1], w[1])), "FETCH_SIZE_KB_avg": round(fk, 1), "WRITE_SIZE_KB_avg": round(wk, 1),
                  "hbm_MB_avg_corrected(2*fetch+write)": round((2 * fk + wk) * 1024 / 1e6, 1)}
    return res


if __name__ == "__main__":
    src, out = sys.argv[1], sys.argv[2]
    res = fold(src)
    json.dump(res, open(out, "w"), indent=1)
    print(f"{len(res)} kernels -> {out}")
