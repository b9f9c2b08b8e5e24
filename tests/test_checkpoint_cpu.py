"""CPU tier: the native checkpoint loader (csrc/loader.hip) -- config.json / generation_config.json parsing and the
safetensors mmap reader -- through the C ABI, against the `safetensors` package and the Python config mirror.  No GPU:
these entry points are host-only (reference: qwen3/generate.rs:22-36, utils/mod.rs:121-137)."""
import ctypes as C
import json
import os
import struct

import numpy as np
import pytest
import torch

from aha_amd import _lib
from aha_amd._lib import AhaHipError
from aha_amd.checkpoint import config_from_desc, config_json, open_weights, parse_config, save_checkpoint
from aha_amd.configs import qwen3_0_6b, qwen3vl_8b, tiny_qwen3, tiny_qwen3_asr, tiny_qwen3vl
from aha_amd.model import make_desc
from aha_amd.weights import qwen3_asr_weights, qwen3_text_weights, qwen3vl_weights


def desc_fields(d):
    out = {}
    for name, _ in d._fields_:
        v = getattr(d, name)
        out[name] = list(v) if hasattr(v, "__len__") else v
    return out


@pytest.mark.parametrize("make_cfg", [tiny_qwen3, qwen3_0_6b, tiny_qwen3vl, qwen3vl_8b, tiny_qwen3_asr],
                         ids=["tiny-qwen3", "qwen3-0.6b", "tiny-vl", "qwen3vl-8b", "tiny-asr"])
def test_config_parse_matches_python_mirror(tmp_path, make_cfg):
    cfg = make_cfg()
    save_checkpoint(str(tmp_path), cfg, {"dummy": torch.zeros(1)})
    got, want = desc_fields(parse_config(str(tmp_path))), desc_fields(make_desc(cfg))
    assert got == want
    assert config_from_desc(parse_config(str(tmp_path))) is not None


def test_eos_scalar_and_unicode_and_errors(tmp_path):
    cfg = tiny_qwen3()
    save_checkpoint(str(tmp_path), cfg, {"dummy": torch.zeros(1)})
    g = json.load(open(tmp_path / "generation_config.json"))
    g["eos_token_id"] = 7
    g["note"] = "café \U0001F600 \\ \" / tab\there"  # escapes the parser must get through
    json.dump(g, open(tmp_path / "generation_config.json", "w"), ensure_ascii=True)
    d = parse_config(str(tmp_path))
    assert d.n_stop_tokens == 1 and d.stop_tokens[0] == 7
    c = json.load(open(tmp_path / "config.json"))
    del c["head_dim"]
    json.dump(c, open(tmp_path / "config.json", "w"))
    with pytest.raises(AhaHipError, match="head_dim"):
        parse_config(str(tmp_path))
    open(tmp_path / "config.json", "w").write('{"hidden_size": 12,')
    with pytest.raises(AhaHipError, match="config.json"):
        parse_config(str(tmp_path))
    with pytest.raises(AhaHipError, match="cannot read"):
        parse_config(str(tmp_path / "nope"))


@pytest.mark.parametrize("bad", ["1e3", "2e0", "12.5", "2.0", "-4", "4294967296"])
def test_config_integers_are_validated_like_serde(tmp_path, bad):
    """usize / u32 fields: a fraction, a non-integral exponent form, a negative or an over-wide value is a parse error in the
    reference (serde_json: any literal with a fraction or exponent is a float), never a silent truncation (strtoll("1e3") == 1) or a later division by zero."""
    cfg = tiny_qwen3()
    save_checkpoint(str(tmp_path), cfg, {"dummy": torch.zeros(1)})
    txt = open(tmp_path / "config.json").read()
    c = json.loads(txt)
    marker = f'"num_key_value_heads": {c["num_key_value_heads"]}'
    assert marker in txt
    open(tmp_path / "config.json", "w").write(txt.replace(marker, f'"num_key_value_heads": {bad}'))
    with pytest.raises(AhaHipError, match="num_key_value_heads"):
        parse_config(str(tmp_path))


@pytest.mark.parametrize("shards", [1, 3])
def test_safetensors_reader_bit_exact(tmp_path, shards):
    cfg = tiny_qwen3vl()
    w = qwen3vl_weights(cfg, seed=5)
    w["extra.f32"] = torch.randn(3, 5, 7)
    w["extra.f16"] = torch.randn(11).half()
    w["extra.i64"] = torch.arange(6).reshape(2, 3)          # listed with dtype -1, never handed to the model
    w["extra.scalar"] = torch.tensor(2.5)                   # 0-d
    w["extra.empty"] = torch.zeros(0, 4, dtype=torch.bfloat16)
    save_checkpoint(str(tmp_path), cfg, w, shards=shards)
    got = open_weights(str(tmp_path))
    assert set(got) == set(w)
    code = {torch.bfloat16: _lib.AHA_BF16, torch.float16: _lib.AHA_F16, torch.float32: _lib.AHA_F32, torch.int64: -1}
    for name, t in w.items():
        dt, shape, raw = got[name]
        assert dt == code[t.dtype] and shape == tuple(t.shape), name
        if dt >= 0:
            want = t.contiguous().reshape(-1).view(torch.uint8).numpy().tobytes() if t.numel() else b""
            assert raw == want, name


def test_safetensors_rejects_corrupt_files(tmp_path):
    d = tmp_path / "a"
    d.mkdir()
    with pytest.raises(AhaHipError, match="no \\*.safetensors"):
        open_weights(str(d))
    hdr = json.dumps({"t": {"dtype": "F32", "shape": [4], "data_offsets": [0, 16]}}).encode()
    (d / "x.safetensors").write_bytes(struct.pack("<Q", len(hdr)) + hdr + b"\0" * 8)      # data shorter than offsets
    with pytest.raises(AhaHipError, match="data_offsets"):
        open_weights(str(d))
    hdr = json.dumps({"t": {"dtype": "F32", "shape": [3], "data_offsets": [0, 16]}}).encode()
    (d / "x.safetensors").write_bytes(struct.pack("<Q", len(hdr)) + hdr + b"\0" * 16)     # shape x dtype != extent
    with pytest.raises(AhaHipError, match="data_offsets"):
        open_weights(str(d))
    (d / "x.safetensors").write_bytes(struct.pack("<Q", 1 << 40) + b"{}")                 # header longer than file
    with pytest.raises(AhaHipError, match="header length"):
        open_weights(str(d))
    (d / "x.safetensors").write_bytes(b"abc")
    with pytest.raises(AhaHipError, match="8-byte"):
        open_weights(str(d))
    good = json.dumps({"t": {"dtype": "F32", "shape": [1], "data_offsets": [0, 4]}}).encode()
    (d / "x.safetensors").write_bytes(struct.pack("<Q", len(good)) + good + b"\0" * 4)
    (d / "y.safetensors").write_bytes(struct.pack("<Q", len(good)) + good + b"\0" * 4)    # same name in two shards
    with pytest.raises(AhaHipError, match="more than one file"):
        open_weights(str(d))


def test_safetensors_reader_survives_odd_names_and_metadata(tmp_path):
    """Tensor names are JSON object keys: escapes, non-ASCII, very long names, a large __metadata__ block, many tensors."""
    from safetensors.torch import save_file
    names = ["plain", "with space", "quote\"inside", "back\\slash", "tab\tname", "unicode-é中\U0001F600", "a" * 300,
             "model.layers.0.self_attn.q_proj.weight", "/slashes/and.dots", "new\nline"]
    names += [f"bulk.{i}" for i in range(400)]
    g = torch.Generator().manual_seed(0)
    tensors = {n: torch.randn((i % 5 + 1, 3), generator=g).to(torch.bfloat16) for i, n in enumerate(names)}
    d = tmp_path / "w"
    d.mkdir()
    save_file(tensors, str(d / "model.safetensors"), metadata={"format": "pt", "note": "x" * 5000, "k\"ey": "v\\al"})
    got = open_weights(str(d))
    assert set(got) == set(tensors)
    for n, t in tensors.items():
        dt, shape, raw = got[n]
        assert dt == _lib.AHA_BF16 and shape == tuple(t.shape)
        assert raw == t.contiguous().reshape(-1).view(torch.uint8).numpy().tobytes()


def test_config_torch_dtype_is_the_string_the_reference_resolves(tmp_path, hip_lib):
    """aha_hip_config_torch_dtype: Qwen3 reads config.json "torch_dtype" (qwen3/config.rs:23), Qwen3-VL "text_config.dtype"
    (qwen3vl/config.rs:100), Qwen3-ASR is the constant "bfloat16" (qwen3_asr/config.rs:186); a host feeds it to aha_hip_get_dtype and
    aha_hip_check_dtype, so an f16 checkpoint without an explicit dtype is refused instead of silently computing in bf16."""
    import ctypes as C
    import json
    from aha_amd import _lib
    buf = C.create_string_buffer(64)

    def write(d, sub):
        p = tmp_path / sub
        p.mkdir()
        (p / "config.json").write_text(json.dumps(d))
        return str(p).encode()

    d = write({"torch_dtype": "float16", "hidden_size": 8}, "q3")
    assert hip_lib.aha_hip_config_torch_dtype(d, buf, 64) == 0 and buf.value == b"float16"
    out = C.c_int32()
    assert hip_lib.aha_hip_get_dtype(-1, buf.value, C.byref(out)) == 0 and out.value == _lib.AHA_F16
    assert hip_lib.aha_hip_check_dtype(out.value) == -6        # AHA_ERR_UNSUPPORTED: refused, not silently bf16
    d = write({"vision_config": {}, "text_config": {"dtype": "bfloat16"}, "torch_dtype": "float32"}, "vl")
    assert hip_lib.aha_hip_config_torch_dtype(d, buf, 64) == 0 and buf.value == b"bfloat16"
    d = write({"thinker_config": {"dtype": "float16"}}, "asr")
    assert hip_lib.aha_hip_config_torch_dtype(d, buf, 64) == 0 and buf.value == b"bfloat16"
    d = write({"hidden_size": 8}, "missing")
    assert hip_lib.aha_hip_config_torch_dtype(d, buf, 64) < 0 and b"torch_dtype" in hip_lib.aha_hip_last_error()
    d = write({"torch_dtype": "bfloat16"}, "small")
    assert hip_lib.aha_hip_config_torch_dtype(d, buf, 4) < 0
