"""Op-level entry points of the C ABI on torch device tensors (torch is only the allocator / stream provider here).

Each function is one kernel family of SURVEY.md section 8a; the parity tests compare them with oracle/ on the same
seeded inputs.  All tensors must live on the GPU and be contiguous bf16 unless stated otherwise.
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from . import _lib
from ._lib import check, lib


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda or not t.is_contiguous():
            raise ValueError("op inputs must be contiguous GPU tensors")


def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    _chk(x, w)
    y = torch.empty_like(x)
    rows = x.numel() // x.shape[-1]
    check(lib().aha_hip_rmsnorm(_ptr(x), _ptr(w), _ptr(y), rows, x.shape[-1], eps, _stream()))
    return y


def gemv(W: torch.Tensor, x: torch.Tensor, norm_w: Optional[torch.Tensor] = None, eps: float = 1e-6,
         residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    _chk(W, x, norm_w, residual)
    N, K = W.shape
    y = torch.empty(N, dtype=torch.bfloat16, device=x.device)
    check(lib().aha_hip_gemv(_ptr(W), _ptr(x), _ptr(y), N, K, _ptr(norm_w), eps, _ptr(residual), _stream()))
    return y


def gemv_gate_up(Wg: torch.Tensor, Wu: torch.Tensor, x: torch.Tensor, norm_w: Optional[torch.Tensor] = None,
                 eps: float = 1e-6) -> torch.Tensor:
    _chk(Wg, Wu, x, norm_w)
    I, K = Wg.shape
    y = torch.empty(I, dtype=torch.bfloat16, device=x.device)
    check(lib().aha_hip_gemv_gate_up(_ptr(Wg), _ptr(Wu), _ptr(x), _ptr(y), I, K, _ptr(norm_w), eps, _stream()))
    return y


def gemm(A: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor] = None,
         residual: Optional[torch.Tensor] = None, act: int = _lib.ACT_NONE) -> torch.Tensor:
    _chk(A, W, bias, residual)
    M, K = A.shape
    N = W.shape[0]
    n_out = N // 2 if act == _lib.ACT_SILU_MUL_PAIRS else N
    Cm = torch.empty(M, n_out, dtype=torch.bfloat16, device=A.device)
    check(lib().aha_hip_gemm(_ptr(A), _ptr(W), _ptr(Cm), M, N, K, K, W.shape[1], n_out, _ptr(bias), _ptr(residual),
                             act, _stream()))
    return Cm


def poison_lds(seed: int) -> None:
    """Test tool: seeded garbage into the LDS of every CU, on the current stream (aha_hip_debug_poison_lds)."""
    check(lib().aha_hip_debug_poison_lds(int(seed) & 0xFFFFFFFF, _stream()))


def gemm_plan(tile: int = 0, splitk: int = 0) -> None:
    """Test hook: force the 128 / 256 tile kernel and a split-K factor for the following GEMMs; (0, 0) = automatic."""
    check(lib().aha_hip_debug_gemm_plan(tile, splitk))


def gemm_grouped(A: torch.Tensor, W: torch.Tensor, Cm: torch.Tensor, M: int, groups: int, a_gstride: int, c_gstride: int, c_row0: int,
                 m_total: int, act: int = _lib.ACT_NONE) -> None:
    """Test entry (aha_hip_debug_gemm_grouped): segment g = A rows [g * a_gstride, + M) -> Cm rows [c_row0 + g * c_gstride, + M) < m_total."""
    _chk(A, W, Cm)
    N, K = W.shape
    check(lib().aha_hip_debug_gemm_grouped(_ptr(A), _ptr(W), _ptr(Cm), M, N, K, Cm.shape[1], act, groups, a_gstride, c_gstride, c_row0, m_total,
                                           _stream()))


def attn_variant(smx: int = -1) -> None:
    """Test hook: the prefill attention's score-chain variant (aha_hip_debug_attn_variant); -1 = default."""
    check(lib().aha_hip_debug_attn_variant(smx))


def attn_form(form: int = -1) -> None:
    """Test hook: the prefill attention's kernel form (aha_hip_debug_attn_form): 16, 64, 65 (64 pipelined) or -1 = automatic."""
    check(lib().aha_hip_debug_attn_form(form))


def interleave_gate_up(Wg: torch.Tensor, Wu: torch.Tensor) -> torch.Tensor:
    """The model loader's fused layout: 16-row blocks alternating gate / up (csrc/model.hip upload_gate_up)."""
    I, K = Wg.shape
    return torch.stack([Wg.reshape(I // 16, 16, K), Wu.reshape(I // 16, 16, K)], dim=1).reshape(2 * I, K).contiguous()


def qknorm_rope(qkv: torch.Tensor, q_norm_w: torch.Tensor, k_norm_w: torch.Tensor, pos: torch.Tensor,
                axis_map: torch.Tensor, nh: int, kvh: int, d: int, eps: float, theta: float):
    """qkv (S, (nh+2kvh)*d) bf16; pos (3,S) int32; axis_map (d/2) int32 -> q (S,nh*d), k (S,kvh*d), v (S,kvh*d)."""
    _chk(qkv, q_norm_w, k_norm_w, pos, axis_map)
    S = qkv.shape[0]
    q = torch.empty(S, nh * d, dtype=torch.bfloat16, device=qkv.device)
    k = torch.empty(S, kvh * d, dtype=torch.bfloat16, device=qkv.device)
    v = torch.empty(S, kvh * d, dtype=torch.bfloat16, device=qkv.device)
    check(lib().aha_hip_qknorm_rope(_ptr(qkv), _ptr(q_norm_w), _ptr(k_norm_w), _ptr(pos), _ptr(axis_map), _ptr(q),
                                    _ptr(k), _ptr(v), S, nh, kvh, d, eps, theta, _stream()))
    return q, k, v


def attn_decode(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, nh: int, kvh: int, d: int,
                scale: Optional[float] = None) -> torch.Tensor:
    """q (nh*d); k, v (L, kvh*d) token-major -> o (nh*d)."""
    _chk(q, k, v)
    L = k.shape[0]
    scale = bf16_scale(d) if scale is None else scale
    o = torch.empty(nh * d, dtype=torch.bfloat16, device=q.device)
    check(lib().aha_hip_attn_decode(_ptr(q), _ptr(k), _ptr(v), _ptr(o), nh, kvh, d, L, scale, _stream()))
    return o


def attn_prefill(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, nh: int, kvh: int, d: int, kv_offset: int = 0,
                 causal: bool = True, scale: Optional[float] = None) -> torch.Tensor:
    """q (S, nh*d); k, v (L, kvh*d) token-major, L = kv_offset + S when causal -> o (S, nh*d)."""
    _chk(q, k, v)
    S, L = q.shape[0], k.shape[0]
    scale = bf16_scale(d) if scale is None else scale
    o = torch.empty(S, nh * d, dtype=torch.bfloat16, device=q.device)
    check(lib().aha_hip_attn_prefill(_ptr(q), _ptr(k), _ptr(v), _ptr(o), S, L, nh, kvh, d, kv_offset, int(causal),
                                     scale, _stream()))
    return o


def argmax(x: torch.Tensor) -> int:
    _chk(x)
    assert x.dtype == torch.float32
    out = torch.zeros(1, dtype=torch.int32, device=x.device)
    check(lib().aha_hip_argmax(_ptr(x), x.numel(), _ptr(out), _stream()))
    return int(out.item()) & 0xFFFFFFFF


def bf16_scale(d: int) -> float:
    """1/sqrt(d) rounded to bf16: the scalar of Candle's `attn_weights * scaling` affine op is cast to the tensor dtype."""
    return float(torch.tensor(1.0 / math.sqrt(d), dtype=torch.float32).bfloat16().float())


def image_to_patches(img_u8_hwc: torch.Tensor, patch: int = 16, merge: int = 2, mean=(0.5, 0.5, 0.5),
                     std=(0.5, 0.5, 0.5)) -> torch.Tensor:
    """V0: (H, W, 3) uint8 GPU tensor -> (N, 3*2*patch*patch) bf16 pixel_values rows (frame duplicated, merge order)."""
    import ctypes as C
    _chk(img_u8_hwc)
    assert img_u8_hwc.dtype == torch.uint8 and img_u8_hwc.dim() == 3 and img_u8_hwc.shape[2] == 3
    H, W = int(img_u8_hwc.shape[0]), int(img_u8_hwc.shape[1])
    out = torch.empty((H // patch) * (W // patch), 6 * patch * patch, dtype=torch.bfloat16, device=img_u8_hwc.device)
    m = (C.c_float * 3)(*mean)
    s = (C.c_float * 3)(*std)
    check(lib().aha_hip_image_to_patches(_ptr(img_u8_hwc), _ptr(out), H, W, patch, merge, m, s, _stream()))
    return out


def video_to_patches(frames_u8_thwc: torch.Tensor, patch: int = 16, merge: int = 2, mean=(0.5, 0.5, 0.5),
                     std=(0.5, 0.5, 0.5)) -> torch.Tensor:
    """process_videos: (T, H, W, 3) uint8 GPU frames -> (ceil(T/2)*N, 3*2*patch*patch) bf16 rows (frame pairs, an odd last frame
    repeated; normalised in bf16 op by op like the reference's video path)."""
    import ctypes as C
    _chk(frames_u8_thwc)
    assert frames_u8_thwc.dtype == torch.uint8 and frames_u8_thwc.dim() == 4 and frames_u8_thwc.shape[3] == 3
    T, H, W = (int(x) for x in frames_u8_thwc.shape[:3])
    out = torch.empty(((T + 1) // 2) * (H // patch) * (W // patch), 6 * patch * patch, dtype=torch.bfloat16, device=frames_u8_thwc.device)
    m = (C.c_float * 3)(*mean)
    s = (C.c_float * 3)(*std)
    check(lib().aha_hip_video_to_patches(_ptr(frames_u8_thwc), _ptr(out), T, H, W, patch, merge, m, s, _stream()))
    return out


def image_resize(img_u8_hwc: torch.Tensor, new_h: int, new_w: int) -> torch.Tensor:
    """V0-pre: DynamicImage::resize_exact(new_w, new_h, CatmullRom) of an RGB8 (H, W, 3) image on the GPU."""
    _chk(img_u8_hwc)
    assert img_u8_hwc.dtype == torch.uint8 and img_u8_hwc.dim() == 3 and img_u8_hwc.shape[2] == 3
    H, W = int(img_u8_hwc.shape[0]), int(img_u8_hwc.shape[1])
    out = torch.empty(new_h, new_w, 3, dtype=torch.uint8, device=img_u8_hwc.device)
    check(lib().aha_hip_image_resize(_ptr(img_u8_hwc), H, W, _ptr(out), new_h, new_w, _stream()))
    return out


def logmel(samples: torch.Tensor) -> torch.Tensor:
    """A0: 16 kHz mono f32 samples on the GPU -> Whisper log-mel features (128, n_samples // 160) f32."""
    _chk(samples)
    assert samples.dtype == torch.float32 and samples.dim() == 1
    out = torch.empty(128, samples.numel() // 160, dtype=torch.float32, device=samples.device)
    check(lib().aha_hip_logmel(_ptr(samples), samples.numel(), _ptr(out), _stream()))
    return out
