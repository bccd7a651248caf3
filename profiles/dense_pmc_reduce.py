"""counter_collection CSVs of the rocprofv3 passes over profiles/dense_pmc_probe.py -> a JSON with the dense kernel's fabric traffic (FETCH_SIZE, doubled on
gfx950 per MI355X_MICROARCH.md's HBM section) against its algorithmic bytes, and its matrix-pipe busy fraction.
    python profiles/dense_pmc_reduce.py <fetch.csv> <mfma.csv> <kernel_stats.csv> <out.json>"""
import csv
import json
import statistics
import sys

KERNEL = "gemm_dense_kernel<5>"


def rows(path):
    with open(path) as f:
        return [r for r in csv.DictReader(f) if KERNEL in r.get("Kernel_Name", r.get("Name", ""))]


def med(rs, counter):
    v = [float(r["Counter_Value"]) for r in rs if r["Counter_Name"] == counter]
    return statistics.median(v) if v else None


fetch = med(rows(sys.argv[1]), "FETCH_SIZE")
mf = rows(sys.argv[2])
busy, cu, gui = med(mf, "SQ_VALU_MFMA_BUSY_CYCLES"), med(mf, "SQ_BUSY_CU_CYCLES"), med(mf, "GRBM_GUI_ACTIVE")
dur = None
for r in csv.DictReader(open(sys.argv[3])):
    if KERNEL in r["Name"]:
        dur = float(r["AverageNs"]) / 1e3
M, N, K = 1408, 28672, 4096
algo = N * K * 2 + M * K * 2 + M * (N // 2) * 2
flop = 2.0 * M * N * K
out = {"kernel": "gemm_dense_kernel<EPI_SWIGLU> 1408 x 28672 x 4096", "avg_us_kernel_trace": dur,
       "tflops": None if dur is None else round(flop / dur / 1e6, 1), "frac_of_2500_dense_peak": None if dur is None else round(flop / dur / 1e6 / 2500, 4),
       "FETCH_SIZE_KiB_raw_median": fetch, "fabric_read_bytes_x2": None if fetch is None else fetch * 2048, "algorithmic_bytes": algo,
       "traffic_over_algorithmic": None if fetch is None else round(fetch * 2048 / algo, 3),
       "SQ_VALU_MFMA_BUSY_CYCLES_median": busy, "SQ_BUSY_CU_CYCLES_median": cu, "GRBM_GUI_ACTIVE_median": gui,
       "note": "SQ_VALU_MFMA_BUSY_CYCLES counts cycles (16 per 16x16x32 bf16 MFMA and SIMD); the ideal for this launch is flop / (1024 flop per cycle and SIMD) = "
               f"{flop / 1024:.3e} SIMD-cycles over 1024 SIMDs"}
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps(out))
