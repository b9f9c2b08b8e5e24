"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's Qwen3-ASR path.

PARITY UNPINNED (see oracle/numerics.py).  Cross-checks in tests/: the log-mel frontend against torch.stft (an
independent STFT) and the audio encoder against HF transformers' Qwen3ASREncoder with the three known deviations of the
reference switched to upstream behaviour.  Reference files restated:
  src/models/feature_extractor/feature_extraction_whisper.rs:93-115     extract_fbank_features
  src/utils/audio_utils.rs:1064-1083 (symmetric Hann), 1157-1301 (Slaney mel filter bank), 1303-1347 (frames x window,
      |rfft|^2), 1483-1503 (extract_frames);  src/utils/tensor_utils.rs:525-549 (pad_reflect_last_dim, incl. its
      right-pad indexing quirk), 354-365 (linspace);  src/models/common/modules.rs:1256-1258 (log10 = ln * 1/ln10)
  src/models/qwen3_asr/model.rs:32-83 (encoder layer), 171-226 (audio encoder forward), 229-366 (thinker)
  src/models/qwen3_asr/processor.rs:187-195 (get_feat_extract_output_lengths)
  src/position_embed/sinusoidal_pe.rs:6-59;  src/models/common/modules.rs:201-242 (NaiveAttention::forward)
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np
import torch

from . import qwen3 as oq
from .numerics import Numerics
from .qwen3vl import gelu_erf, gelu_tanh, layer_norm, linspace_f32


@dataclass
class AsrSwitches:
    """Where the reference deviates from upstream Qwen3-ASR; defaults = the reference's behaviour."""
    global_attention: bool = True      # model.rs:218-220: every layer attends over ALL audio tokens (mask None)
    pe_reference: bool = True          # sinusoidal_pe.rs:13-15: omega_i = 10000^(-i/(d/2)); upstream: 10000^(-i/(d/2-1))
    conv_gelu_tanh: bool = True        # model.rs:200-202 `.gelu()` = Candle's tanh approximation [unverified]; upstream erf


# ---- A0: Whisper log-mel frontend ------------------------------------------------------------------------------------
def create_hann_window(n: int) -> np.ndarray:
    """audio_utils.rs:1064-1083: symmetric Hann in f64, stored f32: 0.5 + 0.5 cos(pi * i / (N-1)), i = 1-N, 3-N, .."""
    i = np.arange(1 - n, n, 2, dtype=np.float64)
    return (0.5 + 0.5 * np.cos(np.pi * i / (n - 1.0))).astype(np.float32)


def hertz_to_mel_slaney(f: np.float32) -> np.float32:
    f = np.float32(f)
    mels = np.float32(3.0) * f / np.float32(200.0)
    if f >= np.float32(1000.0):
        logstep = np.float32(27.0) / np.log(np.float32(6.4))
        mels = np.float32(15.0) + np.log(f / np.float32(1000.0)) * logstep
    return np.float32(mels)


def mel_to_hertz_slaney(m: np.float32) -> np.float32:
    m = np.float32(m)
    freq = np.float32(200.0) * m / np.float32(3.0)
    if m >= np.float32(15.0):
        logstep = np.log(np.float32(6.4)) / np.float32(27.0)
        freq = np.float32(1000.0) * np.exp(logstep * (m - np.float32(15.0)))
    return np.float32(freq)


def mel_filter_bank(n_freq: int = 201, n_mels: int = 128, fmin: float = 0.0, fmax: float = 8000.0,
                    sr: float = 16000.0) -> np.ndarray:
    """audio_utils.rs:1218-1301 (Slaney scale, Slaney norm, triangles in Hz space), all in f32 -> (n_freq, n_mels)."""
    mel_pts = linspace_f32(float(hertz_to_mel_slaney(fmin)), float(hertz_to_mel_slaney(fmax)), n_mels + 2)
    filt = np.array([mel_to_hertz_slaney(m) for m in mel_pts], dtype=np.float32)
    fft = linspace_f32(0.0, sr / 2.0, n_freq)
    diff = (filt[1:] - filt[:-1]).astype(np.float32)
    slopes = (filt[None, :] - fft[:, None]).astype(np.float32)
    down = (np.float32(-1.0) * slopes[:, :-2] / diff[:-1]).astype(np.float32)
    up = (slopes[:, 2:] / diff[1:]).astype(np.float32)
    fb = np.maximum(np.minimum(down, up), np.float32(0.0))
    enorm = (np.float32(2.0) / (filt[2:n_mels + 2] - filt[:n_mels])).astype(np.float32)
    return (fb * enorm[None, :]).astype(np.float32)


def pad_reflect_last_dim(x: np.ndarray, pad_l: int, pad_r: int) -> np.ndarray:
    """tensor_utils.rs:525-549, including its quirk: the right pad is cut from the ALREADY left-padded tensor at the
    un-padded index last_dim - pad_r, i.e. it mirrors original samples [L - pad_r - pad_l, L - pad_l)."""
    L = x.shape[-1]
    t = x
    if pad_l:
        t = np.concatenate([t[1:1 + pad_l][::-1], t])
    if pad_r:
        t = np.concatenate([t, t[L - pad_r:L][::-1]])
    return t


def log_mel(wave: np.ndarray, n_fft: int = 400, hop: int = 160, n_mels: int = 128) -> np.ndarray:
    """extract_fbank_features (feature_extraction_whisper.rs:93-115), dither 0 -> (n_mels, n_frames-1) f32."""
    y = pad_reflect_last_dim(np.asarray(wave, dtype=np.float32), n_fft // 2, n_fft // 2)
    n_frames = 1 + (len(y) - n_fft) // hop
    idx = np.arange(n_fft)[None, :] + hop * np.arange(n_frames)[:, None]
    frames = (y[idx] * create_hann_window(n_fft)[None, :]).astype(np.float32)
    spec = np.fft.rfft(frames.astype(np.float64), axis=-1)          # realfft R2C; |.|^2 (norm_sqr)
    power = (spec.real ** 2 + spec.imag ** 2).astype(np.float32)[: n_frames - 1]   # last frame dropped
    mel = (mel_filter_bank(n_fft // 2 + 1, n_mels).T.astype(np.float64) @ power.T.astype(np.float64)).astype(np.float32)
    mel = np.maximum(mel, np.float32(1e-10))
    lg = (np.log(mel) * np.float32(1.0 / math.log(10.0))).astype(np.float32)
    lg = np.maximum(lg, lg.max() - np.float32(8.0))
    return ((lg + np.float32(4.0)) * np.float32(0.25)).astype(np.float32)


def get_feat_extract_output_lengths(n: int) -> int:
    """processor.rs:187-195."""
    r = n % 100
    if r > 0:
        f = (r - 1) // 2 + 1
        return ((f - 1) // 2 + 1 - 1) // 2 + 1 + (n // 100) * 13
    return (n // 100) * 13


# ---- A1/A2: audio encoder ----------------------------------------------------------------------------------------------
class OracleAudioEncoder:
    """Qwen3ASRAudioEncoder (qwen3_asr/model.rs:85-227)."""

    def __init__(self, acfg, weights: Dict[str, torch.Tensor], nm: Numerics, prefix="thinker.audio_tower.",
                 sw: Optional[AsrSwitches] = None):
        self.c, self.nm, self.p, self.sw = acfg, nm, prefix, sw or AsrSwitches()
        self.w = {k: nm.r(v.float()) for k, v in weights.items() if k.startswith(prefix)}

    def _conv(self, x, name):
        nm = self.nm
        y = torch.nn.functional.conv2d(x.double(), self.w[self.p + name + ".weight"].double(), None, stride=2, padding=1).float()
        y = nm.r(nm.r(y) + self.w[self.p + name + ".bias"].reshape(1, -1, 1, 1))     # conv, then bias broadcast_add
        return nm.r(gelu_tanh(y) if self.sw.conv_gelu_tanh else gelu_erf(y))

    def pos_embed(self, n: int, d: int) -> torch.Tensor:
        half = d // 2
        if self.sw.pe_reference:
            inv = oq.compute_default_rope_parameters(d, 10000.0)                      # 10000^(-i/(d/2)), i < d/2
        else:
            inv = torch.exp(-math.log(10000.0) / (half - 1) * torch.arange(half).float())
        fr = torch.arange(n, dtype=torch.float32)[:, None] * inv[None]
        return torch.cat([fr.sin(), fr.cos()], -1)

    def layer(self, li, x, seg):
        nm, c, p = self.nm, self.c, f"{self.p}layers.{li}."
        g = lambda n: self.w[p + n]
        h = layer_norm(nm, x, g("self_attn_layer_norm.weight"), g("self_attn_layer_norm.bias"), 1e-5)
        n, d = h.shape
        nh, hd = c.encoder_attention_heads, c.d_model // c.encoder_attention_heads
        q = nm.linear(h, g("self_attn.q_proj.weight"), g("self_attn.q_proj.bias")).reshape(n, nh, hd)
        k = nm.linear(h, g("self_attn.k_proj.weight"), g("self_attn.k_proj.bias")).reshape(n, nh, hd)
        v = nm.linear(h, g("self_attn.v_proj.weight"), g("self_attn.v_proj.bias")).reshape(n, nh, hd)
        scale = oq.attn_scale(nm, hd)
        outs = []
        for a, b in zip(seg[:-1], seg[1:]):
            o = oq.eager_attention_forward(nm, q[a:b].transpose(0, 1)[None], k[a:b].transpose(0, 1)[None],
                                           v[a:b].transpose(0, 1)[None], 1, None, scale)
            outs.append(o.reshape(b - a, d))
        o = nm.linear(torch.cat(outs, 0), g("self_attn.out_proj.weight"), g("self_attn.out_proj.bias"))
        x = nm.r(o + x)
        h = layer_norm(nm, x, g("final_layer_norm.weight"), g("final_layer_norm.bias"), 1e-5)
        h = nm.r(gelu_erf(nm.linear(h, g("fc1.weight"), g("fc1.bias"))))          # activation_function "gelu" = erf
        h = nm.linear(h, g("fc2.weight"), g("fc2.bias"))
        return nm.r(h + x)

    def forward(self, feats: torch.Tensor, upto: str = "proj") -> torch.Tensor:
        """feats (n_mels, F) in T -> (n_tokens, output_dim).  model.rs:171-226."""
        nm, c, w = self.nm, self.c, self.w
        F = feats.shape[1]
        win = c.n_window * 2
        lens = [win] * (F // win) + ([F % win] if F % win else [])
        xt = feats.t()
        chunks, s = [], 0
        for L in lens:
            ch = xt[s:s + L]
            if L < win:
                ch = torch.cat([ch, torch.zeros(win - L, xt.shape[1])], 0)
            chunks.append(ch)
            s += L
        x = torch.stack(chunks, 0).transpose(1, 2).unsqueeze(1)                     # (C,1,mels,win)
        n_tok = sum(get_feat_extract_output_lengths(L) for L in lens)
        x = self._conv(self._conv(self._conv(x, "conv2d1"), "conv2d2"), "conv2d3")
        b, ch, f, t = x.shape
        x = x.permute(0, 3, 1, 2).reshape(b, t, ch * f)
        x = nm.linear(x, w[self.p + "conv_out.weight"])
        x = nm.r(x + nm.r(self.pos_embed(t, c.d_model))[None])
        x = x.reshape(b * t, -1)[:n_tok]
        if self.sw.global_attention:
            seg = [0, n_tok]
        else:   # upstream: windows of n_window_infer / (2 n_window) chunks
            per = (c.n_window_infer // win) * get_feat_extract_output_lengths(win)
            seg = list(range(0, n_tok, per)) + [n_tok]
        for li in range(c.encoder_layers):
            x = self.layer(li, x, seg)
        x = layer_norm(nm, x, w[self.p + "ln_post.weight"], w[self.p + "ln_post.bias"], 1e-5)
        if upto == "ln_post":
            return x
        x = nm.r(gelu_erf(nm.linear(x, w[self.p + "proj1.weight"], w[self.p + "proj1.bias"])))
        return nm.linear(x, w[self.p + "proj2.weight"], w[self.p + "proj2.bias"])


class OracleQwen3ASR:
    """Qwen3ASRModel / Qwen3ASRThinker (qwen3_asr/model.rs:308-420): audio tokens scattered into the Qwen3 text tower.
    The ASR M-RoPE variant with identical T/H/W rows (position_ids None, model.rs:265-275) is plain 1-D RoPE."""

    def __init__(self, cfg, weights, nm: Optional[Numerics] = None, sw: Optional[AsrSwitches] = None):
        self.cfg, self.nm = cfg, nm or Numerics()
        self.text = oq.OracleQwen3(cfg.text, weights, self.nm, prefix="thinker.model.", lm_head_name="thinker.lm_head.weight")
        self.audio = OracleAudioEncoder(cfg.audio, weights, self.nm, sw=sw)
        self.last_audio_embeds = None

    def clear_cache(self):
        self.text.clear_cache()

    def stop_token_ids(self):
        return self.text.stop_token_ids()

    def forward(self, input_ids, seqlen_offset, input_features: Optional[torch.Tensor] = None):
        t = self.text
        ids = list(input_ids)
        x = t.embed_tokens(ids)
        if input_features is not None:
            af = self.audio.forward(self.nm.r(torch.as_tensor(input_features).float()))
            rows = [i for i, tok in enumerate(ids) if tok == self.cfg.audio_token_id]
            if len(rows) != af.shape[0]:
                raise ValueError(f"n_audio_tokens num: {len(rows)} not equal to audio_feature len: {af.shape[0]}")
            x = x.clone()
            x[0, rows] = af
            self.last_audio_embeds = af
        h = t.forward_hidden(None, x, seqlen_offset)
        return self.nm.linear(h, t.lm_head)

    def forward_initial(self, input_ids, seqlen_offset, mm=None):
        return self.forward(input_ids, seqlen_offset, mm)

    def forward_step(self, input_ids, seqlen_offset):
        return self.forward(input_ids, seqlen_offset, None)
