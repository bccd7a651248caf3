"""Per-kernel roofline table of a bench trace: algorithmic bytes (or flops) of every major kernel of the step / its average rocprofv3 duration.
usage: python3 profiles/kernel_rooflines.py profiles/r04   (reads bench_kernel_stats_prof{1,64,128,64x4}.csv)
Algorithmic bytes: the weights of the projection (bf16, read once) for the weight-streaming GEMMs, the cached K and V rows for the decode attention
(streams x 1006 keys x 8 kv heads x 128 dims x 2 B x 2), flops = 2 M N K for the prefill GEMMs.  Peaks: 8 TB/s, 2.5 PFLOP/s dense bf16."""
import csv, re, sys
D = sys.argv[1] if len(sys.argv) > 1 else "profiles/r04"
MB = 1e6
W = {"qkv": 6144 * 4096 * 2, "o": 4096 * 4096 * 2, "gateup": 28672 * 4096 * 2, "down": 4096 * 14336 * 2, "lm_head": 128263 * 4096 * 2}
KV1 = 1006 * 8 * 128 * 2 * 2  # one stream's cached K + V of one layer


def stats(tag):
    out = {}
    for r in csv.DictReader(open(f"{D}/bench_kernel_stats_prof{tag}.csv")):
        out[r["Name"]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
    return out


def find(st, pat):
    hits = [(n, v) for n, v in st.items() if re.search(pat, n)]
    return max(hits, key=lambda h: h[1][0]) if hits else None


def line(st, label, pat, nbytes=None, flops=None):
    h = find(st, pat)
    if not h:
        return
    name, (calls, us) = h
    if nbytes is not None:
        rate = nbytes / us / 1e6
        print(f"  {label:44s} {us:8.2f} us  {nbytes / MB:8.1f} MB  {rate:5.2f} TB/s  {rate / 8:5.2f} of 8 TB/s   ({calls} launches)")
    else:
        rate = flops / us / 1e9
        print(f"  {label:44s} {us:8.2f} us  {flops / 1e9:8.1f} GF  {rate:5.2f} PF/s  {rate / 2.5:5.2f} of 2.5 PF/s ({calls} launches)")


s = stats("1")
print("one stream (configs[1]), decode pass:")
line(s, "gate/up GEMV (fused norm, self-paired tiles)", r"gemm_skinny_kernel<1, 1, 8", W["gateup"])
line(s, "down_proj GEMV", r"gemm_skinny_kernel<1, 1, 3", W["down"])
line(s, "q/k/v GEMV (fused norm)", r"gemm_skinny_kernel<1, 1, 0, true, 2", W["qkv"])
line(s, "attention + combine + o_proj (one launch)", r"llm_attn_oproj_kernel", W["o"] + KV1)
line(s, "lm_head GEMV", r"gemm_skinny_kernel<1, 1, 6, true, 2", W["lm_head"])
s = stats("64")
print("64 streams (configs[2]), decode pass (64 rows) and prefill (1408 rows):")
line(s, "gate/up (gemm_wide, norm on stage)", r"gemm_wide_kernel<4, 4, 6, 5, true", W["gateup"])
line(s, "q/k/v (gemm_mid)", r"gemm_mid_kernel<4, 2, 7, 1, true", W["qkv"])
line(s, "o_proj + down_proj avg (gemm_mid)", r"gemm_mid_kernel<4, 2, 7, 1, false", (W["o"] + W["down"]) / 2)
line(s, "decode attention", r"llm_attn_partial_kernel<4, 1, true", 64 * KV1)
line(s, "lm_head (gemm_wide)", r"gemm_wide_kernel<4, 4, 6, 6, true", W["lm_head"])
line(s, "prefill gate/up (gemm_dense)", r"gemm_dense_kernel<5>", flops=2.0 * 1408 * 28672 * 4096)
line(s, "prefill q/k/v (gemm_dense)", r"gemm_dense_kernel<0>", flops=2.0 * 1408 * 6144 * 4096)
line(s, "prefill o_proj + down + encoder out/fc2 avg (gemm_dense, K slices)", r"gemm_dense_kernel<7>", flops=2.0 * (160 * 1408 * 4096 * (4096 + 14336) + 120 * 3072 * 1024 * (1024 + 4096)) / 560)
line(s, "encoder fc1 at 3072 rows (gemm_dense)", r"gemm_dense_kernel<2>", flops=2.0 * 3072 * 4096 * 1024)
line(s, "encoder attention, 64 streams", r"enc_attention_kernel<3", 64 * 16 * 624 * 64 * 2 * 2)
line(s, "prefill attention", r"llm_attn_prefill_kernel", 64 * KV1)
s = stats("128")
print("128 streams, decode pass (128 rows):")
line(s, "gate/up (gemm_wide)", r"gemm_wide_kernel<8, 4, 6, 5", W["gateup"])
line(s, "q/k/v + o_proj + down avg (gemm_wide)", r"gemm_wide_kernel<8, 4, 6, 7", (W["qkv"] + W["o"] + W["down"]) / 3)
line(s, "decode attention", r"llm_attn_partial_kernel<4, 1, true", 128 * KV1)
s = stats("64x4")
print("64 streams x beam 4 (the reference's production decoding), decode pass (256 rows):")
line(s, "gate/up (gemm_wide)", r"gemm_wide_kernel<16, 2, 4, 5", W["gateup"])
line(s, "gate/up (gemm_wide), as flops", r"gemm_wide_kernel<16, 2, 4, 5", flops=2.0 * 256 * 28672 * 4096)
line(s, "q/k/v + down avg (gemm_wide, one 256-row wg)", r"gemm_wide_kernel<16, 2, 4, 7", (W["qkv"] + W["down"]) / 2)
line(s, "q/k/v + down avg, as flops", r"gemm_wide_kernel<16, 2, 4, 7", flops=2.0 * 256 * (6144 * 4096 + 4096 * 14336) / 2)
line(s, "o_proj (gemm_wide, two 128-row wgs)", r"gemm_wide_kernel<8, 4, 6, 7", W["o"])
line(s, "decode attention (shared prefix, folded)", r"llm_attn_partial_kernel<4, 1, true", 64 * KV1)
