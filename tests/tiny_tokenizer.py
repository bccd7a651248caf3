"""A REAL (tiny) Hugging Face tokenizer directory for the toy model, so that the agent's `load_model(args)` can be driven exactly
as SimulEval drives the reference: `transformers.AutoTokenizer.from_pretrained(args.model_name, use_fast=False)`, `add_tokens`,
`apply_chat_template`, `decode`.  Word-level vocabulary of 1024 entries whose special ids equal `toy_config()`'s; the chat template
has Llama-3.1's shape: a system turn that always opens with a fixed "date" block, so that a conversation WITHOUT a system message
starts with BOS + system header + 20 content tokens = 25 tokens before the first <|eot_id|> -- the 25 the reference strips from
later chunks (agents/infinisst.py:264).  Also writes config.json / generation_config.json with the toy Llama's side parameters."""
import json
import os

DATE_BLOCK = "Cutting Knowledge Date: December 2023 Today Date: 26 Jul 2024 f1 f2 f3 f4 f5 f6 f7 f8 f9\n\n"  # 19 words + "\n\n" = 20 tokens

CHAT_TEMPLATE = (
    "{{- bos_token }}"
    "{%- if messages[0]['role'] == 'system' %}{%- set system_message = messages[0]['content'] %}{%- set messages = messages[1:] %}"
    "{%- else %}{%- set system_message = '' %}{%- endif %}"
    "{{- '<|start_header_id|>system<|end_header_id|>\n\n' }}{{- '" + DATE_BLOCK.replace("\n", "\\n") + "' }}{{- system_message }}{{- '<|eot_id|>' }}"
    "{%- for message in messages %}{{- '<|start_header_id|>' + message['role'] + '<|end_header_id|>\n\n' + message['content'] + '<|eot_id|>' }}{%- endfor %}"
    "{%- if add_generation_prompt %}{{- '<|start_header_id|>assistant<|end_header_id|>\n\n' }}{%- endif %}"
)


def build_tokenizer_dir(path, cfg, name="toy-Llama-3.1-tiny"):
    """Writes <path>/<name>/{tokenizer.json, tokenizer_config.json, config.json, generation_config.json}; returns the directory
    (its name contains "3.1": the reference picks the 25-token strip from the model name, agents/infinisst.py:183)."""
    from tokenizers import AddedToken, Regex, Tokenizer, models, pre_tokenizers
    d = os.path.join(str(path), name)
    os.makedirs(d, exist_ok=True)
    base = cfg.sp_patch_id  # ids below this are the "pretrained" vocabulary; preprocess() adds the speech / latency tokens after it
    names = {cfg.bos_id: "<|begin_of_text|>", cfg.start_header_id: "<|start_header_id|>", cfg.end_header_id: "<|end_header_id|>",
             cfg.eot_id: "<|eot_id|>", cfg.pad_id: "<|finetune_right_pad_id|>", cfg.user_id: "user", cfg.assistant_id: "assistant",
             cfg.system_id: "system", cfg.nl2_id: "\n\n", 7: "(x", 0: "<unk>"}
    for e in cfg.eos_ids:
        names.setdefault(e, f"<|eos_{e}|>")
    words = ("Translate the following speech from English to German with latency . Cutting Knowledge Date: December 2023 Today 26 Jul "
             "2024 f1 f2 f3 f4 f5 f6 f7 f8 f9").split()
    free = [i for i in range(8, base) if i not in names]
    for w, i in zip(dict.fromkeys(words), free):
        names[i] = w
    vocab = {names.get(i, f"w{i}"): i for i in range(base)}
    tok = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.Split(Regex(" +"), behavior="removed")
    special = sorted(i for i, n in names.items() if n.startswith("<|"))
    tok.add_special_tokens([AddedToken(names[i], special=True, normalized=False) for i in special])
    tok.add_tokens([AddedToken("\n\n", normalized=False)])
    tok.save(os.path.join(d, "tokenizer.json"))
    with open(os.path.join(d, "tokenizer_config.json"), "w") as f:
        json.dump({"tokenizer_class": "PreTrainedTokenizerFast", "bos_token": "<|begin_of_text|>", "eos_token": "<|eot_id|>",
                   "chat_template": CHAT_TEMPLATE, "model_max_length": 131072, "clean_up_tokenization_spaces": False}, f)
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump({"model_type": "llama", "hidden_size": cfg.llm_dim, "intermediate_size": cfg.llm_ffn, "num_hidden_layers": cfg.llm_layers,
                   "num_attention_heads": cfg.llm_heads, "num_key_value_heads": cfg.llm_kv_heads, "vocab_size": base, "max_position_embeddings": 131072,
                   "rms_norm_eps": cfg.rms_eps, "rope_theta": cfg.rope_theta,
                   "rope_scaling": {"rope_type": "llama3", "factor": cfg.rope_factor, "low_freq_factor": cfg.rope_low_freq_factor,
                                    "high_freq_factor": cfg.rope_high_freq_factor,
                                    "original_max_position_embeddings": cfg.rope_original_max_pos}}, f)
    with open(os.path.join(d, "generation_config.json"), "w") as f:
        json.dump({"bos_token_id": cfg.bos_id, "eos_token_id": list(cfg.eos_ids)}, f)
    return d
