"""Builds profiles/*_pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE):
    make_pmc_json.py fetch_counter_collection.csv write_counter_collection.csv out.json "<command profiled>"
HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KiB -> bytes (the x2 on FETCH_SIZE is the gfx950
correction of MI355X_MICROARCH.md for 16-B-per-lane streams)."""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_sources_sha():          # same definition as bench.py's
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(ROOT, "markov-process-analysis-on-point-cloud_amd", "csrc", "*.h*"))):
        h.update(os.path.basename(path).encode())
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


ENTRY = {          # C-ABI entry point -> (main kernel prefix, helper kernel prefixes)
    "mpa_gemm_f32/tiled": ("gemm_kernel<", ["splitk_reduce_kernel"]),
    "mpa_gemm_f32/shortk": ("gemm_shortk_kernel<", []),
    "mpa_gemm_tn_grouped_f32": ("gemm_tn_grouped_kernel", ["splitk_reduce_grouped_kernel"]),
    "mpa_gemm_grouped_f32": ("gemm_nt_grouped_kernel<", []),
    "mpa_gemm_grouped_bf16": ("gemm_bf16_grouped_kernel<", []),
    "mpa_knn_f32": ("knn_mfma_kernel<", []),
    "mpa_diffattn_fwd_f32": ("diffattn_fwd", []),
    "mpa_diffattn_bwd_f32": ("diffattn_bwd_p1", ["diffattn_bwd_p2", "csr_build_kernel"]),
    "mpa_gemm_bf16": ("gemm_bf16_kernel<", []),
    "mpa_gemm_tn_grouped_bf16": ("gemm_bf16_tn_grouped_kernel", ["splitk_reduce_grouped_kernel"]),
    "mpa_diffattn_fwd_bf16": ("diffattn_fwd", []),
    "mpa_diffattn_bwd_bf16": ("diffattn_bwd_p1", ["diffattn_bwd_p2", "csr_build_kernel"]),
    "mpa_fps_f32": ("fps_kernel<", []),
    "mpa_bn_act_fwd_f32": ("bn_act_fwd_kernel", []),
}


def load(path):
    tot, cnt = collections.Counter(), collections.Counter()
    seen = set()
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if k.startswith("_ZN12_GLOBAL__N_1"):          # a mangled name rocprofv3 left as is: <len><name>...
            rest = k[len("_ZN12_GLOBAL__N_1"):]
            n = int("".join(c for c in rest[:3] if c.isdigit()))
            k = rest[len(str(n)):len(str(n)) + n] + "<"
        tot[k] += float(r["Counter_Value"])
        key = (k, r.get("Dispatch_Id"))
        if key not in seen:
            seen.add(key)
            cnt[k] += 1
    return tot, cnt


fetch, fcnt = load(sys.argv[1])
write, wcnt = load(sys.argv[2])
out = {"how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `%s` (MI355X); counter unit "
              "KiB; per-launch averages over every dispatch of the run; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
              "-- the x2 on FETCH_SIZE is the gfx950 correction of MI355X_MICROARCH.md (calibrated for 16-B-per-lane "
              "streams; uncalibrated for narrower loads)." % sys.argv[4],
       "kernel_sources_sha": kernel_sources_sha(), "kernels": {}, "by_device_kernel": {}}
for k in sorted(fetch, key=lambda k: -(2 * fetch[k] + write.get(k, 0))):
    n = max(fcnt[k], 1)
    out["by_device_kernel"][k] = {"launches": fcnt[k], "fetch_kib_per_launch": fetch[k] / n,
                                  "write_kib_per_launch": write.get(k, 0) / max(wcnt.get(k, 0), 1),
                                  "hbm_bytes_per_launch": (2 * fetch[k] / n + write.get(k, 0) / max(wcnt.get(k, 0), 1)) * 1024}
for entry, (main, helpers) in ENTRY.items():
    mains = [k for k in fetch if k.startswith(main)]
    n = sum(fcnt[k] for k in mains)
    if not n:
        continue
    ks = mains + [k for k in fetch for h in helpers if k.startswith(h)]
    fb = sum(fetch[k] for k in ks)
    wb = sum(write.get(k, 0) * fcnt[k] / max(wcnt.get(k, 0), 1) for k in ks)
    out["kernels"][entry] = {"launches": n, "device_kernels": sorted(ks), "fetch_kib": fb / n, "write_kib": wb / n,
                             "hbm_bytes_per_launch": (2 * fb + wb) / n * 1024}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
