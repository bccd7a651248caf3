"""Toy language model shared by tests/golden/gen_golden.py::gen_beam_loop (which drives the REFERENCE's beam-search loop on it)
and tests/test_oracle_golden.py (which drives oracle/beam.py::beam_search_loop on it)."""
import torch


def toy_beam_forward(E, O, bias, decay):
    """The toy language model of beam_loop.npz: the 'KV cache' of a beam is the list of embedding rows E[token] of every token
    it has consumed; logits = tanh(sum_i decay^(n-1-i) * cache[i]) @ O + bias.  Sensitive to the WHOLE cache content and order,
    so a wrong reorder / clone / expansion changes the scores.  Used by the generator (batched, below) and by
    tests/test_oracle_golden.py (one beam at a time, through oracle/beam.py::beam_search_loop)."""
    def forward(tokens, kv, first):
        for t in tokens:
            kv.append(E[int(t)].clone())
        n = len(kv)
        wts = decay ** torch.arange(n - 1, -1, -1, dtype=torch.float32)
        h = (torch.stack(kv) * wts[:, None]).sum(0)
        return torch.tanh(h) @ O + bias
    return forward
