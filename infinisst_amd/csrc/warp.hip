// Sampling branch of generate (host side): HF's logits WARPERS and the multinomial draw.
//
// Reference: agents/infinisst.py:311-315 passes do_sample / top_p / top_k / epsilon_cutoff / temperature into the patched generate, whose
// sample branch (model/patches/patch_hf.py:606-624) runs HF `_sample` [3P transformers 4.47.0]: the processors (device: sample.hip), then the warpers
// `_get_logits_processor` appends for do_sample -- Temperature -> TopK -> TopP -> Epsilon, in that order (recorded from the image's transformers 5.15:
// tests/golden/sampling_warpers.npz) -- then softmax + torch.multinomial.  None of the reference's scripts enables it, so it is not a hot path: the
// processed scores of a step come to the host (0.5 MB per stream and step next to a 15 GB weight pass) and this file does the rest in fp32, op for op
// as the warpers do (kept sets are pinned to HF's classes by the fixture).  torch.multinomial's stream of random numbers cannot be reproduced; the draw
// here is the inverse CDF of the warped distribution (vocabulary order) at a uniform from a counter-based generator keyed by
// (seed, stream, chunk, step): reproducible, independent of batching, restated identically in oracle/generate.py.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

#include "common.h"
#include "kernels.h"

static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
double sample_uniform(uint64_t seed, int stream, int chunk, int step) {
    uint64_t x = splitmix64(seed);
    x = splitmix64(x ^ (uint64_t)(uint32_t)stream);
    x = splitmix64(x ^ ((uint64_t)(uint32_t)chunk << 20));
    x = splitmix64(x ^ (uint64_t)(uint32_t)step);
    return (double)(x >> 11) * (1.0 / 9007199254740992.0);  // 53 bits -> [0, 1)
}

static void softmax_f32(const float* x, int n, std::vector<float>& p) {
    float m = -std::numeric_limits<float>::infinity();
    for (int i = 0; i < n; ++i) m = std::max(m, x[i]);
    p.resize(n);
    double sum = 0.0;
    for (int i = 0; i < n; ++i) { p[i] = std::isinf(x[i]) && x[i] < 0 ? 0.f : std::exp(x[i] - m); sum += p[i]; }
    const float inv = (float)(1.0 / sum);
    for (int i = 0; i < n; ++i) p[i] *= inv;
}

// scores: processed fp32 scores of one row, warped in place (-inf = removed): Temperature -> TopK -> TopP -> Epsilon
void warp_scores(float* s, int n, float temperature, int top_k, float top_p, float epsilon, int min_keep) {
    const float NEG = -std::numeric_limits<float>::infinity();
    min_keep = std::max(1, std::min(min_keep, n));  // min_tokens_to_keep: 1 for the sample branch, n_eos + 1 under beam search (HF _get_logits_processor)
    if (temperature > 0.f && temperature != 1.0f)  // TemperatureLogitsWarper
        for (int i = 0; i < n; ++i) s[i] = s[i] / temperature;
    if (top_k > 0) {  // TopKLogitsWarper: remove everything below the k-th largest score (ties with it stay)
        const int k = std::min(std::max(top_k, min_keep), n);
        std::vector<float> c(s, s + n);
        std::nth_element(c.begin(), c.begin() + (k - 1), c.end(), std::greater<float>());
        const float kth = c[k - 1];
        for (int i = 0; i < n; ++i) if (s[i] < kth) s[i] = NEG;
    }
    if (top_p < 1.0f) {  // TopPLogitsWarper: ascending sort, remove the head whose cumulative probability is <= 1 - top_p, keep at least one
        std::vector<int> idx(n);
        for (int i = 0; i < n; ++i) idx[i] = i;
        std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return s[a] < s[b]; });
        std::vector<float> sorted(n), p;
        for (int i = 0; i < n; ++i) sorted[i] = s[idx[i]];
        softmax_f32(sorted.data(), n, p);
        float cum = 0.f;
        const float lim = 1.0f - top_p;
        for (int i = 0; i < n - min_keep; ++i) {  // (the last = most probable min_keep tokens always stay)
            cum += p[i];
            if (cum <= lim) s[idx[i]] = NEG;
        }
    }
    if (epsilon > 0.f && epsilon < 1.0f) {  // EpsilonLogitsWarper: remove probabilities below epsilon, never the most probable token
        std::vector<float> p;
        softmax_f32(s, n, p);
        float top = NEG;  // the min_keep-th largest score: nothing at or above it is removed
        if (min_keep <= 1) {
            for (int i = 0; i < n; ++i) top = std::max(top, s[i]);
        } else {
            std::vector<float> c(s, s + n);
            std::nth_element(c.begin(), c.begin() + (min_keep - 1), c.end(), std::greater<float>());
            top = c[min_keep - 1];
        }
        for (int i = 0; i < n; ++i) if (p[i] < epsilon && s[i] < top) s[i] = NEG;
    }
}

// warpers + softmax + one multinomial draw (inverse CDF in vocabulary order).  Returns the drawn token.
int warp_and_sample(float* s, int n, float temperature, int top_k, float top_p, float epsilon, double u) {
    warp_scores(s, n, temperature, top_k, top_p, epsilon, 1);
    std::vector<float> p;
    softmax_f32(s, n, p);
    double total = 0.0;
    for (int i = 0; i < n; ++i) total += p[i];
    const double target = u * total;
    double cum = 0.0;
    int last = 0;
    for (int i = 0; i < n; ++i) {
        if (p[i] <= 0.f) continue;
        cum += p[i];
        last = i;
        if (cum > target) return i;
    }
    return last;
}

// Beam sample (patch_hf.py:871-875): `torch.multinomial(softmax(scores), k)` without replacement over the flattened scores of all beams, as k sequential
// inverse-CDF draws in index order at the given uniforms, each over what is left; fp32 softmax, fp64 running sums, left to right (oracle/generate.py
// multinomial_without_replacement: the same arithmetic).  Returns ISST_ERR_STATE when fewer than k entries have non-zero probability (torch raises there).
int multinomial_without_replacement(const float* scores, long n, int k, const double* u, long* picked) {
    float m = -std::numeric_limits<float>::infinity();
    for (long i = 0; i < n; ++i) m = std::max(m, scores[i]);
    if (!(m > -std::numeric_limits<float>::infinity())) return ISST_ERR_STATE;
    std::vector<float> pf((size_t)n);
    double sum = 0.0;
    for (long i = 0; i < n; ++i) { pf[i] = (std::isinf(scores[i]) && scores[i] < 0) ? 0.f : std::exp(scores[i] - m); sum += pf[i]; }
    const float inv = (float)(1.0 / sum);
    std::vector<double> p((size_t)n);
    for (long i = 0; i < n; ++i) p[i] = (double)(pf[i] * inv);
    for (int j = 0; j < k; ++j) {
        double total = 0.0;
        for (long i = 0; i < n; ++i) total += p[i];
        if (!(total > 0.0)) return ISST_ERR_STATE;
        const double target = u[j] * total;
        double cum = 0.0;
        long pick = -1, last = -1;
        for (long i = 0; i < n; ++i) {
            if (p[i] <= 0.0) continue;
            cum += p[i];
            last = i;
            if (cum > target) { pick = i; break; }
        }
        if (pick < 0) pick = last;
        picked[j] = pick;
        p[pick] = 0.0;
    }
    return ISST_OK;
}

extern "C" int isst_op_multinomial_wor(const float* scores, long n, int k, const double* uniforms, long* picked) {
    if (!scores || n < 1 || k < 1 || !uniforms || !picked) return ISST_ERR_ARG;
    for (int j = 0; j < k; ++j) if (!(uniforms[j] >= 0.0 && uniforms[j] < 1.0)) return ISST_ERR_ARG;
    return multinomial_without_replacement(scores, n, k, uniforms, picked);
}
extern "C" int isst_op_warp(float* scores, int vocab, float temperature, int top_k, float top_p, float epsilon_cutoff, int min_tokens_to_keep) {
    if (!scores || vocab < 1 || min_tokens_to_keep < 1) return ISST_ERR_ARG;
    warp_scores(scores, vocab, temperature, top_k, top_p, epsilon_cutoff, min_tokens_to_keep);
    return ISST_OK;
}
extern "C" int isst_op_warp_sample(float* scores, int vocab, float temperature, int top_k, float top_p, float epsilon_cutoff, double u, int* token) {
    if (!scores || vocab < 1 || !token || !(u >= 0.0 && u < 1.0)) return ISST_ERR_ARG;
    *token = warp_and_sample(scores, vocab, temperature, top_k, top_p, epsilon_cutoff, u);
    return ISST_OK;
}
extern "C" double isst_op_sample_uniform(uint64_t seed, int stream, int chunk, int step) { return sample_uniform(seed, stream, chunk, step); }
