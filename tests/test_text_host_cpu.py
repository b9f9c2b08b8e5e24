"""CPU tier: host text plumbing -- TokenizerModel (tokenizer/mod.rs:10-121) and ChatTemplate
(chat_template/mod.rs:7-160) mirrors.  No checkpoint files are on disk and there is no network, so the fixtures are built here:
a small byte-level BPE trained with the `tokenizers` wheel (the same crate version the reference pins, Cargo.lock:4637) and a
chat template written in the shape of Qwen3's that exercises every construct fix_template rewrites."""
import json

import pytest

from aha_amd import text_host as th

tokenizers = pytest.importorskip("tokenizers")

CORPUS = ["hello world, this is a tiny corpus for a byte level bpe", "the quick brown fox jumps over the lazy dog",
          "line one\nline two\n\nline four", "naïve café déjà vu — ünïcödé", "<|im_start|>user\nhi<|im_end|>\n<|im_start|>assistant\n"] * 4


@pytest.fixture(scope="module")
def tok_dir(tmp_path_factory):
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers, trainers
    d = tmp_path_factory.mktemp("tok")
    tok = Tokenizer(models.BPE())
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False, use_regex=False)
    tok.decoder = decoders.ByteLevel()
    tr = trainers.BpeTrainer(vocab_size=400, initial_alphabet=pre_tokenizers.ByteLevel.alphabet(), special_tokens=[])
    tok.train_from_iterator(CORPUS, tr)
    tok.model.save(str(d))                                    # vocab.json + merges.txt
    base = tok.get_vocab_size()
    json.dump({"added_tokens_decoder": {str(base): {"content": "<|im_start|>", "special": True},
                                        str(base + 1): {"content": "<|im_end|>", "special": True},
                                        str(base + 2): {"content": "<tool_call>", "special": False}},
               "chat_template": TEMPLATE}, open(d / "tokenizer_config.json", "w"))
    return d


TEMPLATE = (
    "{%- if tools %}{{- '<|im_start|>system\\n# Tools\\n' }}{%- for tool in tools %}{{- '\\n' }}{{- tool | tojson }}{%- endfor %}{{- '<|im_end|>\\n' }}{%- endif %}"
    "{%- for message in messages %}"
    "{%- if message.content is string %}{%- set content = message.content %}{%- else %}{%- set content = '' %}{%- endif %}"
    "{%- if message.role == 'user' and not (content.startswith('<tool_response>') and content.endswith('</tool_response>')) %}"
    "{{- '<|im_start|>user\\n' + content + '<|im_end|>\\n' }}"
    "{%- elif message.role == 'assistant' %}"
    "{%- set reasoning_content = '' %}"
    "{%- if '</think>' in content %}"
    "{%- set reasoning_content = content.split('</think>')[0].rstrip('\\n').split('<think>')[-1].lstrip('\\n') %}"
    "{%- set content = content.split('</think>')[-1].lstrip('\\n') %}"
    "{%- endif %}"
    "{{- '<|im_start|>assistant\\n' + content + '<|im_end|>\\n' }}"
    "{%- else %}{{- '<|im_start|>' + message.role + '\\n' + content.lstrip('\\n') + '<|im_end|>\\n' }}"
    "{%- endif %}"
    "{%- endfor %}"
    "{%- if add_generation_prompt %}{{- '<|im_start|>assistant\\n' }}{%- if enable_thinking is defined and enable_thinking is false %}{{- '<think>\\n\\n</think>\\n\\n' }}{%- endif %}{%- endif %}\n")


def test_tokenizer_both_init_paths_agree(tok_dir, tmp_path):
    """vocab.json + merges.txt + added_tokens_decoder (mod.rs:28-86) builds the same tokenizer a saved tokenizer.json gives."""
    a = th.TokenizerModel.init(str(tok_dir))
    full = tmp_path / "full"
    full.mkdir()
    a.tokenizer.save(str(full / "tokenizer.json"))
    b = th.TokenizerModel.init(str(full))
    for text in CORPUS[:5] + ["", " leading space", "tabs\tand\nnewlines\n", "<|im_start|>system\nx<|im_end|>", "emoji \U0001F600 ok"]:
        ia, ib = a.text_encode_vec(text, True), b.text_encode_vec(text, True)
        assert ia == ib
        assert a.token_decode_with_special(ia) == text                 # byte-level BPE round-trips every string
    ids = a.text_encode("<|im_start|>user\nhi<|im_end|>")
    assert a.token_decode(ids) == "user\nhi"                           # skip_special_tokens = true drops the special ones ...
    # every added_tokens_decoder entry goes through add_special_tokens (mod.rs:80), so "special": false ones are skipped too
    assert a.token_decode(a.text_encode("a<tool_call>b")) == "ab" and a.token_decode_with_special(a.text_encode("a<tool_call>b")) == "a<tool_call>b"
    with pytest.raises(FileNotFoundError, match="vocab.json"):
        th.TokenizerModel.init(str(tmp_path))


def test_stream_decode_hook_holds_back_split_utf8(tok_dir):
    """What the streaming loop relies on (generate.rs:198-214): a multi-byte character split across two tokens decodes to U+FFFD
    until both are present."""
    t = th.TokenizerModel.init(str(tok_dir))
    ids = t.text_encode_vec("\U0001F600", False)
    assert len(ids) >= 2 and "�" in t.token_decode(ids[:1]) and t.token_decode(ids) == "\U0001F600"


def test_fix_template_rewrites_and_lookup_order(tok_dir, tmp_path):
    fixed = th.get_template(str(tok_dir))
    assert "startswith" not in fixed and "is startingwith('<tool_response>')" in fixed and "is endingwith('</tool_response>')" in fixed
    assert "((content | split('</think>'))[0] | rstrip('\\n') | split('<think>'))[-1] | lstrip('\\n')" in fixed
    assert "(content | split('</think>'))[-1] | lstrip('\\n')" in fixed and "content | lstrip('\\n')" in fixed
    assert th.fix_template("a{%- generation -%}b{%- endgeneration -%}c") == "abc"
    # lookup order: tokenizer_config.json -> chat_template.json -> chat_template.jinja; nothing without tokenizer_config.json
    d = tmp_path / "m"
    d.mkdir()
    (d / "chat_template.jinja").write_text("J")
    with pytest.raises(ValueError, match="chat_template is none"):
        th.get_template(str(d))
    (d / "tokenizer_config.json").write_text("{}")
    assert th.get_template(str(d)) == "J"
    (d / "chat_template.json").write_text(json.dumps({"chat_template": "C"}))
    assert th.get_template(str(d)) == "C"
    (d / "tokenizer_config.json").write_text(json.dumps({"chat_template": "T"}))
    assert th.get_template(str(d)) == "T"


def test_rewritten_template_renders_like_the_python_original(tok_dir):
    """The reference renders the REWRITTEN template with minijinja + its filters; the original template is Python-flavoured jinja.
    Both must give the same prompt: rendered here with jinja2 twice (original with str methods, rewritten with the mirror's filters)."""
    import jinja2
    msgs = [{"role": "system", "content": "\n\nYou are terse."},
            {"role": "user", "content": "What is 2+2?"},
            {"role": "assistant", "content": "<think>\nsimple\n</think>\n\n4"},
            {"role": "user", "content": "<tool_response>\n{\"x\": 1}\n</tool_response>"},
            {"role": "user", "content": "thanks"}]
    tools = [{"type": "function", "function": {"name": "f", "parameters": {"a": 1}}}]
    ct = th.ChatTemplate.init(str(tok_dir))
    env = jinja2.Environment()   # keep_trailing_newline = false: minijinja's default (the reference never changes it)
    env.filters["tojson"] = lambda v: json.dumps(v, ensure_ascii=False, separators=(",", ":"))
    orig = env.from_string(TEMPLATE)
    for tl in (None, tools):
        for et in (None, False, True):
            got = ct.apply_chat_template(msgs, tl, et)
            want = orig.render(messages=msgs, tools=tl, add_generation_prompt=True, enable_thinking=bool(et))
            assert got == want
            tail = "<|im_start|>assistant\n" + ("" if et else "<think>\n\n</think>\n\n")
            assert got.endswith(tail + ("\n" if TEMPLATE.endswith("\n\n") else ""))   # ONE trailing newline of the source is dropped
    # enable_thinking = metadata OR request flag (mod.rs:142-146); metadata parses like str::parse::<bool>
    assert ct.apply_chat_template(msgs, None, None, {"enable_thinking": "true"}) == ct.apply_chat_template(msgs, None, True)
    assert ct.apply_chat_template(msgs, None, None, {"enable_thinking": "True"}) == ct.apply_chat_template(msgs, None, False)
    # end to end: template -> tokenizer -> ids -> text
    t = th.TokenizerModel.init(str(tok_dir))
    prompt = ct.apply_chat_template(msgs[1:2])
    ids = t.text_encode(prompt)
    assert t.token_decode_with_special(ids) == prompt and ids[0] == t.tokenizer.token_to_id("<|im_start|>")


def test_rust_style_strip_filters():
    """lstrip / rstrip with an argument are str::trim_start_matches / trim_end_matches of the WHOLE pattern (mod.rs:103-111), not
    Python's character-set strip."""
    ct = th.ChatTemplate.str_init("{{ x | lstrip('ab') }}|{{ x | rstrip('ab') }}|{{ ' y ' | lstrip }}|{{ v | string }}")
    assert ct.template.render(x="ababbaab", v=True) == "baab|ababba|y |true"
    # format!("{}", v) of minijinja's none / undefined is "none" (chat_template/mod.rs string filter)
    assert th.ChatTemplate.str_init("{{ v | string }}|{{ missing | string }}").template.render(v=None) == "none|"


def test_trailing_newline_of_the_template_source_is_dropped():
    """minijinja's Environment::new() keeps keep_trailing_newline = false and the reference never sets it (chat_template/mod.rs:84-139):
    a template file ending in a newline renders WITHOUT it (one newline only).  Expected strings are hard-coded, not rendered."""
    ct = th.ChatTemplate.str_init("{% for m in messages %}<{{ m.role }}>{{ m.content }}\n{% endfor %}A:\n")
    assert ct.apply_chat_template([{"role": "user", "content": "hi"}]) == "<user>hi\nA:"
    ct = th.ChatTemplate.str_init("A:\n\n")
    assert ct.apply_chat_template([]) == "A:\n"
    ct = th.ChatTemplate.str_init("A:")
    assert ct.apply_chat_template([]) == "A:"
