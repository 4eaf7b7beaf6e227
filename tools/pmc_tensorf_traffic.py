"""HBM traffic of the TensoRF colour factor backward (s3d_vm_color_backward = k_vm_bound + plane + line + k_vm_flush_reduce) from two
rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only beside --pmc) over tools/bench_tensorf_step.py:
    python tools/pmc_tensorf_traffic.py <fetch_dir> <write_dir> > profiles/rNN_tensorf_pmc.json
KiB counters; FETCH_SIZE x 2 on gfx950 (MI355X_MICROARCH.md, HBM section).  k_vm_bound and k_vm_flush_reduce serve both backward
calls of a step (density, colour) under one name: their dispatches alternate, the colour call's are the half with the larger mean."""
import collections, csv, glob, hashlib, json, os, sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_dispatch(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter and "k_vm_" in r["Kernel_Name"]:
            out[r["Kernel_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    return {k: [v for _, v in sorted(vs)] for k, vs in out.items()}


def pick(table, key, colour_half=False):
    for name, vals in table.items():
        if key in name:
            if colour_half:
                a, b = vals[0::2], vals[1::2]
                vals = a if sum(a) / max(len(a), 1) >= sum(b) / max(len(b), 1) else b
            return sum(vals) / len(vals), len(vals)
    return 0.0, 0


fetch, write = per_dispatch(sys.argv[1], "FETCH_SIZE"), per_dispatch(sys.argv[2], "WRITE_SIZE")
rows, total = {}, 0.0
for key, half in (("k_vm_bound", True), ("k_vm_plane_backward_mm<3", False), ("k_vm_line_backward_mm<3", False), ("k_vm_flush_reduce", True)):
    f, n = pick(fetch, key, half)
    w, _ = pick(write, key, half)
    rows[key] = {"dispatches": n, "fetch_kib_raw": f, "fetch_x2_kib": 2 * f, "write_kib": w}
    total += (2 * f + w) * 1024
src = open(os.path.join(REPO, "seal-3d_amd", "csrc", "tensorf.hip"), "rb").read()
print(json.dumps({"tensorf_hip_sha256_16": hashlib.sha256(src).hexdigest()[:16], "color_backward_bytes_per_launch": total,
                  "kernels": rows, "command": "tools/bench_tensorf_step.py 300 fused native (105,051 samples per step)",
                  "note": "FETCH_SIZE x 2 + WRITE_SIZE, KiB x 1,024, per launch of the colour call's kernels"}, indent=1))
