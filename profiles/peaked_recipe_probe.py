"""How decisive is greedy decoding under a synthetic weight recipe?  (VERDICT r02 weak #1.)

CPU only.  A mid-size decoder (full vocabulary, narrower / shallower than Llama-3.1-8B so that fp32 + bf16 copies fit in this
container) runs several decode steps in fp32 and in bf16 from the same bf16-representable weights; reported per recipe:
sigma of the logits, the bf16 noise (|bf16 - fp32|, mean / max), the fp32 top-2 margin AFTER the logits processors are out of the
way (raw logits) and the fraction of steps whose margin exceeds 2.5 x the max noise ("decisive").

    python profiles/peaked_recipe_probe.py [dim] [layers]
"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from infinisst_amd import synth  # noqa: E402
from infinisst_amd.config import full_config  # noqa: E402
from oracle import llm as ollm  # noqa: E402

dim = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 16
torch.set_num_threads(8)
cfg = full_config().replace(llm_dim=dim, llm_heads=dim // 128, llm_kv_heads=max(1, dim // 512), llm_ffn=int(3.5 * dim), llm_layers=layers)


def llm_weights(recipe):
    g = torch.Generator().manual_seed(1)
    w = {}
    for name, shape in synth.weight_shapes(cfg).items():
        if not (name.startswith("model.layers.") or name in ("model.embed_tokens.weight", "model.norm.weight", "lm_head.weight")):
            continue
        if name.endswith("norm.weight"):
            w[name] = torch.ones(shape)
        else:
            w[name] = 0.02 * torch.randn(shape, generator=g)
    synth.apply_recipe(cfg, w, recipe)
    return {k: v.bfloat16() for k, v in w.items()}


for recipe in sys.argv[3:] or ["plain", "peaked"]:
    w16 = llm_weights(recipe)
    w32 = {k: v.float() for k, v in w16.items()}
    rope16, rope32 = ollm.llm_rope_tables(cfg, 512, torch.bfloat16), ollm.llm_rope_tables(cfg, 512, torch.float32)
    g = torch.Generator().manual_seed(5)
    prompt = synth.chunk_prompt_ids(cfg, 1, first=True)
    speech = (0.5 * torch.randn(12, dim, generator=g)).bfloat16()
    kv16, kv32 = ollm.new_kv(cfg), ollm.new_kv(cfg)
    seq = list(prompt)
    rows = []
    t0 = time.time()
    with torch.inference_mode():
        for step in range(14):
            ids = torch.tensor(seq if step == 0 else seq[-1:])
            l32 = ollm.model_forward(w32, cfg, ids, kv32, rope32, speech=speech.float() if step == 0 else None).float()
            l16 = ollm.model_forward(w16, cfg, ids, kv16, rope16, speech=speech if step == 0 else None).float()
            noise = (l16 - l32).abs()
            top = torch.topk(l32, 2)
            margin = float(top.values[0] - top.values[1])
            rows.append((float(l32.std()), float(noise.mean()), float(noise.max()), margin, int(top.indices[0]), int(torch.argmax(l16))))
            seq.append(int(top.indices[0]))
    r = np.array([x[:4] for x in rows])
    dec = np.mean(r[:, 3] > 2.5 * r[:, 2])
    print(f"[{recipe}] dim {dim} layers {layers}: sigma {r[:,0].mean():.3f}  bf16 noise mean {r[:,1].mean():.4f} max {r[:,2].mean():.4f} (worst {r[:,2].max():.4f})  "
          f"fp32 top-2 margin median {np.median(r[:,3]):.3f} min {r[:,3].min():.3f}  margin/max-noise median {np.median(r[:,3]/r[:,2]):.1f}  "
          f"decisive {100*dec:.0f} %  argmax agree {np.mean([x[4]==x[5] for x in rows])*100:.0f} %  tokens {[x[4] for x in rows]}  ({time.time()-t0:.0f} s)", flush=True)
