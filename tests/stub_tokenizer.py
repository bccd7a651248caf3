"""A tokenizer stub with the surface the reference uses (apply_chat_template, decode, eos_token_id, len) and the token
structure of Llama-3.1-Instruct's chat template: without a system message the template injects its default system header
(25 tokens before the closing <|eot_id|>, which is why the reference strips `[:, 25:]`, agents/infinisst.py:264)."""
import re


class StubTokenizer:
    def __init__(self, cfg, n_words=200):
        self.cfg = cfg
        self.bos = 1000 if cfg.vocab < 2000 else 128000
        self.eos_token_id = cfg.eot_id
        self.words = {}
        self.n_words = n_words
        self.special = {"<sp_patch>": cfg.sp_patch_id}

    def __len__(self):
        return self.cfg.vocab

    def _word_id(self, w):
        if w not in self.words:
            self.words[w] = 10 + (len(self.words) % self.n_words)
        return self.words[w]

    def _content(self, text):
        ids = []
        for piece in re.findall(r"<sp_patch>|<latency_\d+>|\S+", text):
            if piece == "<sp_patch>":
                ids.append(self.cfg.sp_patch_id)
            elif piece.startswith("<latency_"):
                ids.append(self.cfg.sp_patch_id + 2 + int(piece[9:-1]))  # <latency_1..4> follow <sp_patch>,<sp_start>,<sp_end>
            else:
                ids.append(self._word_id(piece))
        return ids

    def apply_chat_template(self, batch, **kw):
        c = self.cfg
        out = []
        for messages in batch:
            ids = [self.bos]
            if not messages or messages[0]["role"] != "system":
                # default header: BOS + SH system EH \n\n + 20 "date" tokens = 25 tokens, then EOT
                ids += [c.start_header_id, 3, c.end_header_id, c.nl2_id] + [4] * 20 + [c.eot_id]
            for m in messages:
                role = {"system": 3, "user": c.user_id, "assistant": c.assistant_id}[m["role"]]
                ids += [c.start_header_id, role, c.end_header_id, c.nl2_id]
                if m["role"] == "system":
                    ids += [4] * 20  # the date lines sit inside the system turn
                ids += self._content(m["content"]) + [c.eot_id]
            out.append(ids)
        return out

    def decode(self, ids, skip_special_tokens=True):
        if isinstance(ids, int):
            ids = [ids]
        inv = {v: k for k, v in self.words.items()}
        toks = []
        for i in ids:
            if i in inv:
                toks.append(inv[i])
            elif i == 7:
                toks.append("(x")  # a token the non-language scan must catch
            elif not skip_special_tokens or i < 1000:
                toks.append(f"w{i}")
        return " ".join(toks)
