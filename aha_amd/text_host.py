"""Host text plumbing of the request path: the reference's tokenizer wrapper and chat-template
renderer, mirrored one to one.  Nothing here touches the GPU; it feeds `input_ids` to forward_initial and turns tokens back
into text for the streaming loop (`sampling.generate_stream_generic_text`'s `decode` hook).

* `TokenizerModel` == /root/reference/src/tokenizer/mod.rs:10-121.  The reference uses the `tokenizers` crate 0.22.2
  (Cargo.lock:4637); the Python wheel of the SAME crate version is installed here, so encode/decode results are the crate's own.
* `ChatTemplate` == /root/reference/src/chat_template/mod.rs:7-160: the template lookup order (tokenizer_config.json ->
  chat_template.json -> chat_template.jinja), `fix_template`'s textual rewrites of Python string methods into filters/tests
  (minijinja has no str methods), the custom filters (tojson, split, lstrip, rstrip, string) and `apply_chat_template`
  (add_generation_prompt = true, enable_thinking = metadata OR request flag).  Rendered here with jinja2 configured to behave
  like that minijinja environment: the rewritten template and the same filters.
"""
from __future__ import annotations

import json
import os
from typing import Any, Dict, List, Optional, Sequence


class TokenizerModel:
    """tokenizer/mod.rs:10-121."""

    def __init__(self, tokenizer):
        self.tokenizer = tokenizer

    @classmethod
    def init(cls, path: str) -> "TokenizerModel":
        from tokenizers import AddedToken, Tokenizer
        from tokenizers import decoders, models, pre_tokenizers
        assert os.path.exists(path), "model path file not exists"            # mod.rs:21-24 (an assert! in the reference too)
        tokenizer_file = os.path.join(path, "tokenizer.json")
        if os.path.exists(tokenizer_file):
            return cls(Tokenizer.from_file(tokenizer_file))
        vocab_file, merges_file = os.path.join(path, "vocab.json"), os.path.join(path, "merges.txt")
        if not os.path.exists(vocab_file):
            raise FileNotFoundError("Neither tokenizer.json nor vocab.json found in model path")
        if not os.path.exists(merges_file):
            raise FileNotFoundError("Neither tokenizer.json nor merges.txt found in model path")
        tok = Tokenizer(models.BPE.from_file(vocab_file, merges_file))
        tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False, trim_offsets=True, use_regex=False)   # ByteLevel::new(false, true, false)
        tok.decoder = decoders.ByteLevel()
        config_file = os.path.join(path, "tokenizer_config.json")
        if os.path.exists(config_file):
            cfg = json.load(open(config_file, encoding="utf-8"))
            added = cfg.get("added_tokens_decoder")
            if isinstance(added, dict):
                specials = [AddedToken(info["content"], special=bool(info.get("special", False)))
                            for info in added.values() if isinstance(info, dict) and isinstance(info.get("content"), str)]
                if specials:
                    tok.add_special_tokens(specials)
        return cls(tok)

    def text_encode_vec(self, text: str, add_special_token: bool) -> List[int]:
        return list(self.tokenizer.encode(text, add_special_tokens=add_special_token).ids)

    def text_encode(self, text: str) -> List[int]:
        """text_encode(text, device): add_special_tokens = true; the (1, S) tensor is the caller's `input_ids`."""
        return self.text_encode_vec(text, True)

    def token_decode(self, tokens: Sequence[int]) -> str:
        return self.tokenizer.decode(list(tokens), skip_special_tokens=True)

    def token_decode_with_special(self, tokens: Sequence[int]) -> str:
        return self.tokenizer.decode(list(tokens), skip_special_tokens=False)


_FIXES = [   # fix_template, chat_template/mod.rs:7-35, in the reference's order
    ("content.startswith('<tool_response>')", "content is startingwith('<tool_response>')"),
    ("content.endswith('</tool_response>')", "content is endingwith('</tool_response>')"),
    ("content.split('</think>')[0].rstrip('\\n').split('<think>')[-1].lstrip('\\n')",
     "((content | split('</think>'))[0] | rstrip('\\n') | split('<think>'))[-1] | lstrip('\\n')"),
    ("content.split('</think>')[-1].lstrip('\\n')", "(content | split('</think>'))[-1] | lstrip('\\n')"),
    ("reasoning_content.strip('\\n')", "reasoning_content | strip('\\n')"),
    ("content.lstrip('\\n')", "content | lstrip('\\n')"),
    ("{%- generation -%}", ""),
    ("{%- endgeneration -%}", ""),
]


def fix_template(chat_template: str) -> str:
    for a, b in _FIXES:
        chat_template = chat_template.replace(a, b)
    return chat_template


def get_template(path: str) -> str:
    """chat_template/mod.rs:37-82: tokenizer_config.json["chat_template"], else chat_template.json, else chat_template.jinja --
    and, as in the reference, NOTHING is looked up when tokenizer_config.json itself is missing."""
    tcfg = os.path.join(path, "tokenizer_config.json")
    tem = None
    if os.path.exists(tcfg):
        v = json.load(open(tcfg, encoding="utf-8")).get("chat_template")
        tem = v if isinstance(v, str) else None
        if tem is None:
            cj = os.path.join(path, "chat_template.json")
            if os.path.exists(cj):
                v = json.load(open(cj, encoding="utf-8")).get("chat_template")
                tem = v if isinstance(v, str) else None
        if tem is None:
            jp = os.path.join(path, "chat_template.jinja")
            if os.path.exists(jp):
                tem = open(jp, encoding="utf-8").read()
    if tem is None:
        raise ValueError("chat_template is none")
    return fix_template(tem)


def _trim_start_matches(s: str, pat: str) -> str:
    """Rust str::trim_start_matches(&str): strips REPEATED occurrences of the whole pattern (not a character set)."""
    if not pat:
        return s
    while s.startswith(pat):
        s = s[len(pat):]
    return s


def _trim_end_matches(s: str, pat: str) -> str:
    if not pat:
        return s
    while s.endswith(pat):
        s = s[:-len(pat)]
    return s


class ChatTemplate:
    """chat_template/mod.rs:84-160."""

    def __init__(self, fixed_template: str):
        import jinja2
        # minijinja 2.x: Environment::new() has keep_trailing_newline = false and the reference never calls
        # set_keep_trailing_newline (chat_template/mod.rs:84-139), so ONE trailing newline of the template source is dropped --
        # jinja2's default too
        env = jinja2.Environment()
        env.filters["tojson"] = lambda v: json.dumps(v, ensure_ascii=False, separators=(",", ":"))   # serde_json::to_string
        env.filters["split"] = lambda s, d: str(s).split(d)
        env.filters["lstrip"] = lambda s, chars=None: str(s).lstrip() if chars is None else _trim_start_matches(str(s), chars)
        env.filters["rstrip"] = lambda s, chars=None: str(s).rstrip() if chars is None else _trim_end_matches(str(s), chars)
        def _no_strip(*_a, **_k):   # fix_template rewrites `reasoning_content.strip(..)` into a `strip` FILTER that setup_environment never
            raise jinja2.TemplateRuntimeError("unknown filter: filter strip is unknown")   # registers: minijinja fails when the branch runs
        env.filters["strip"] = _no_strip
        # the reference's filter is format!("{}", v): minijinja 2.x's Display for Value writes "none" for None and NOTHING for Undefined
        # ([unverified]: the crate is not on disk; from its published source, ValueRepr::Undefined => Ok(()))
        env.filters["string"] = lambda v: "" if isinstance(v, jinja2.Undefined) else ("none" if v is None else (str(v).lower() if isinstance(v, bool) else str(v)))
        env.tests["startingwith"] = lambda s, p: str(s).startswith(p)
        env.tests["endingwith"] = lambda s, p: str(s).endswith(p)
        self.env = env
        self.template = env.from_string(fixed_template)

    @classmethod
    def init(cls, path: str) -> "ChatTemplate":
        if not os.path.exists(path):
            raise FileNotFoundError("model path not found")
        return cls(get_template(path))

    @classmethod
    def str_init(cls, chat_template: str) -> "ChatTemplate":
        return cls(fix_template(chat_template))

    def apply_chat_template(self, messages: List[Dict[str, Any]], tools: Optional[List[Any]] = None,
                            enable_thinking: Optional[bool] = None, metadata: Optional[Dict[str, str]] = None) -> str:
        """apply_chat_template (mod.rs:141-160): enable_thinking = metadata["enable_thinking"] OR the request's flag (default false);
        add_generation_prompt = true."""
        meta = None
        if metadata is not None and metadata.get("enable_thinking") in ("true", "false"):   # str::parse::<bool>: exactly these two
            meta = metadata["enable_thinking"] == "true"
        et = bool(meta) or bool(enable_thinking)
        return self.template.render(messages=messages, tools=tools, add_generation_prompt=True, enable_thinking=et)
