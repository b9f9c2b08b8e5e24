//! `aha-hip`: the reference-side binding of `include/aha_hip.h`.
//!
//! * [`sys`] -- `extern "C"` declarations and `#[repr(C)]` mirrors of the header's structs (std only).
//! * [`Model`] -- a safe handle: lifecycle, `forward_initial` / `forward_step` / `clear_cache` / `stop_token_ids`, the
//!   device-resident greedy loop with a per-token callback (what a streaming generate loop needs), the sampled path's
//!   candidate query.
//! * feature `aha`: `impl InferenceModel for HipQwen3` (reference `src/models/common/mod.rs:25-45`), i.e. what
//!   `generate_generic` / `generate_stream_generic` (`src/models/common/generate.rs:87-368`) and the ASR loop
//!   (`src/models/qwen3_asr/generate.rs:130-186`) call.
//!
//! Never compiled in the repository's own build image (no Rust toolchain there) -- see Cargo.toml.

pub mod sys {
    use std::ffi::{c_char, c_void};

    #[repr(C)]
    pub struct AhaCtx {
        _p: [u8; 0],
    }
    #[repr(C)]
    pub struct AhaRng {
        _private: [u8; 0],
    }
    #[repr(C)]
    pub struct AhaModel {
        _p: [u8; 0],
    }

    pub const AHA_BF16: i32 = 0;
    pub const AHA_F16: i32 = 1;
    pub const AHA_F32: i32 = 2;
    pub const AHA_ARCH_QWEN3: i32 = 0;
    pub const AHA_ARCH_QWEN3VL: i32 = 1;
    pub const AHA_ARCH_QWEN3ASR: i32 = 2;

    /// `aha_model_desc` (field order and widths checked against the header by tests/test_host_cpu.py on the ctypes mirror)
    #[repr(C)]
    #[derive(Clone, Copy, Default)]
    pub struct AhaModelDesc {
        pub arch: i32,
        pub hidden_size: i32,
        pub intermediate_size: i32,
        pub num_hidden_layers: i32,
        pub num_attention_heads: i32,
        pub num_key_value_heads: i32,
        pub head_dim: i32,
        pub vocab_size: i32,
        pub rms_norm_eps: f32,
        pub rope_theta: f32,
        pub tie_word_embeddings: i32,
        pub mrope_section: [i32; 3],
        pub vis_depth: i32,
        pub vis_hidden_size: i32,
        pub vis_num_heads: i32,
        pub vis_intermediate_size: i32,
        pub vis_in_channels: i32,
        pub vis_patch_size: i32,
        pub vis_temporal_patch_size: i32,
        pub vis_spatial_merge_size: i32,
        pub vis_out_hidden_size: i32,
        pub vis_num_position_embeddings: i32,
        pub vis_deepstack_indexes: [i32; 8],
        pub vis_num_deepstack: i32,
        pub image_token_id: i32,
        pub video_token_id: i32,
        pub vision_start_token_id: i32,
        pub vision_end_token_id: i32,
        pub kv_reserve_tokens: i32,
        pub n_stop_tokens: i32,
        pub stop_tokens: [u32; 8],
        pub aud_d_model: i32,
        pub aud_encoder_layers: i32,
        pub aud_attention_heads: i32,
        pub aud_ffn_dim: i32,
        pub aud_num_mel_bins: i32,
        pub aud_downsample_hidden_size: i32,
        pub aud_output_dim: i32,
        pub aud_n_window: i32,
        pub audio_token_id: i32,
        pub tp_rank: i32,
        pub tp_size: i32,
        /// AHA_BF16 (0) is the only compute dtype; anything else makes `aha_hip_model_create` fail (AHA_ERR_UNSUPPORTED)
        pub compute_dtype: i32,
    }

    /// `aha_tensor_view`
    #[repr(C)]
    pub struct AhaTensorView {
        pub name: *const c_char,
        pub data: *const c_void,
        pub dtype: i32,
        pub ndim: i32,
        pub shape: [i64; 5],
        pub on_device: i32,
    }

    /// `aha_mm_input`
    #[repr(C)]
    pub struct AhaMmInput {
        pub pixel_values: *const c_void,
        pub pixel_dtype: i32,
        pub n_patches: i64,
        pub image_grid_thw: *const u32,
        pub n_images: i32,
        pub audio_features: *const f32,
        pub n_frames: i64,
        pub audio_samples: *const f32,
        pub n_samples: i64,
        pub image_embeds: *const c_void,
        pub n_image_tokens: i64,
        pub pixel_values_video: *const c_void,
        pub n_patches_video: i64,
        pub video_grid_thw: *const u32,
        pub n_videos: i32,
    }
    impl AhaMmInput {
        pub fn empty() -> Self {
            Self {
                pixel_values: std::ptr::null(),
                pixel_dtype: AHA_F32,
                n_patches: 0,
                image_grid_thw: std::ptr::null(),
                n_images: 0,
                audio_features: std::ptr::null(),
                n_frames: 0,
                audio_samples: std::ptr::null(),
                n_samples: 0,
                image_embeds: std::ptr::null(),
                n_image_tokens: 0,
                pixel_values_video: std::ptr::null(),
                n_patches_video: 0,
                video_grid_thw: std::ptr::null(),
                n_videos: 0,
            }
        }
    }

    extern "C" {
        pub fn aha_hip_init(device: i32, out: *mut *mut AhaCtx) -> i32;
        pub fn aha_hip_shutdown(ctx: *mut AhaCtx);
        pub fn aha_hip_last_error() -> *const c_char;
        pub fn aha_hip_version() -> *const c_char;
        pub fn aha_hip_get_dtype(requested: i32, cfg_dtype: *const c_char, out: *mut i32) -> i32;
        pub fn aha_hip_check_dtype(dtype: i32) -> i32;
        pub fn aha_hip_model_create(
            ctx: *mut AhaCtx,
            desc: *const AhaModelDesc,
            weights: *const AhaTensorView,
            n_weights: usize,
            out: *mut *mut AhaModel,
        ) -> i32;
        /// config.json + generation_config.json + every *.safetensors of `dir`, parsed / mmapped by the library itself
        pub fn aha_hip_config_parse(dir: *const c_char, out: *mut AhaModelDesc) -> i32;
        pub fn aha_hip_config_torch_dtype(dir: *const c_char, out: *mut c_char, cap: usize) -> i32;
        pub fn aha_hip_model_load(ctx: *mut AhaCtx, dir: *const c_char, kv_reserve_tokens: usize, out: *mut *mut AhaModel) -> i32;
        pub fn aha_hip_model_destroy(m: *mut AhaModel);
        pub fn aha_hip_forward_initial(
            m: *mut AhaModel,
            ids: *const u32,
            n_ids: usize,
            seqlen_offset: usize,
            mm: *const AhaMmInput,
            logits_out: *mut f32,
            argmax_out: *mut u32,
        ) -> i32;
        pub fn aha_hip_forward_step(m: *mut AhaModel, token: u32, seqlen_offset: usize, logits_out: *mut f32, argmax_out: *mut u32) -> i32;
        pub fn aha_hip_clear_cache(m: *mut AhaModel) -> i32;
        pub fn aha_hip_stop_token_ids(m: *const AhaModel, out: *mut u32, cap: usize) -> i32;
        pub fn aha_hip_decode_greedy(m: *mut AhaModel, first_token: u32, seqlen_offset: usize, max_new: usize, tokens_out: *mut u32) -> i32;
        pub fn aha_hip_sample_candidates(
            m: *mut AhaModel,
            context: *const u32,
            n_context: usize,
            repeat_penalty: f32,
            temperature: f32,
            k: i32,
            vals_out: *mut f32,
            idx_out: *mut u32,
            max_out: *mut f32,
            sumexp_out: *mut f32,
        ) -> i32;
        pub fn aha_hip_last_logits(m: *mut AhaModel, logits_out: *mut f32) -> i32;
        pub fn aha_hip_rng_create(seed: u64, out: *mut *mut AhaRng) -> i32;
        pub fn aha_hip_rng_destroy(rng: *mut AhaRng);
        pub fn aha_hip_rng_next_u32(rng: *mut AhaRng) -> u32;
        pub fn aha_hip_rng_weighted_index(rng: *mut AhaRng, weights: *const f32, n: usize, index_out: *mut u32) -> i32;
        pub fn aha_hip_kv_export(m: *mut AhaModel, out_dev: *mut c_void, out_bytes: usize, bytes_needed: *mut usize, n_tokens: *mut usize, rope_delta: *mut i64) -> i32;
        pub fn aha_hip_kv_import(m: *mut AhaModel, in_dev: *const c_void, in_bytes: usize, src_heads: i32, src_head0: i32, dst_head0: i32, n_heads: i32, n_tokens: usize, rope_delta: i64) -> i32;
        pub fn aha_hip_set_gemm_reserved_cus(n: i32) -> i32;
        /// context-parallel prefill: full weights on every rank, the prompt's rows sharded, one K / V all-gather per layer
        pub fn aha_hip_tp_unique_id(out128: *mut c_void) -> i32;
        pub fn aha_hip_set_context_parallel(
            m: *mut AhaModel,
            rank: i32,
            world: i32,
            all_gather: Option<unsafe extern "C" fn(buf_dev: *mut c_void, bytes_per_rank: usize, user: *mut c_void) -> i32>,
            user: *mut c_void,
        ) -> i32;
        pub fn aha_hip_cp_init_rccl(m: *mut AhaModel, unique_id128: *const c_void) -> i32;
        pub fn aha_hip_embed(m: *mut AhaModel, ids: *const u32, n_ids: usize, out: *mut f32) -> i32;
        pub fn aha_hip_cache_len(m: *const AhaModel) -> usize;
        pub fn aha_hip_audio_resample(
            ctx: *mut AhaCtx,
            pcm: *const f32,
            n_frames: i64,
            channels: i32,
            orig_sr: i32,
            target_sr: i32,
            out: *mut f32,
            out_cap: i64,
        ) -> i64;
    }
}

use std::ffi::{c_char, CStr, CString};
use std::fmt;

/// Non-zero status of the library + its thread-local message (`aha_hip_last_error`)
#[derive(Debug, Clone)]
pub struct Error {
    pub code: i32,
    pub message: String,
}
impl fmt::Display for Error {
    fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
        write!(f, "aha_hip error {}: {}", self.code, self.message)
    }
}
impl std::error::Error for Error {}

fn check(rc: i32) -> Result<(), Error> {
    if rc >= 0 {
        return Ok(());
    }
    let message = unsafe { CStr::from_ptr(sys::aha_hip_last_error()) }.to_string_lossy().into_owned();
    Err(Error { code: rc, message })
}

/// `get_dtype(dtype, cfg_dtype)` of the reference (`src/utils/mod.rs:77-115`) with a `hip` arm: an explicit request wins,
/// otherwise the checkpoint's `torch_dtype`; bfloat16 stays bfloat16 (gfx950 computes in it natively).
pub fn get_dtype(requested: Option<i32>, cfg_dtype: &str) -> Result<i32, Error> {
    let c = CString::new(cfg_dtype).unwrap_or_default();
    let mut out = 0i32;
    check(unsafe { sys::aha_hip_get_dtype(requested.unwrap_or(-1), c.as_ptr(), &mut out) })?;
    Ok(out)
}

/// Multi-modal payload of `forward_initial` (host memory, borrowed for the call).
pub enum MmInput<'a> {
    None,
    /// Qwen3-VL: processor output `(n_patches, 1536)` f32 + `image_grid_thw` `(n_images, 3)` (qwen3vl/generate.rs:79-101)
    Image { pixel_values: &'a [f32], n_patches: usize, grid_thw: &'a [u32] },
    /// Qwen3-VL with videos: `data_vec[0..4]` = pixel_values, image_grid_thw, pixel_values_video, video_grid_thw, each pair
    /// optional (qwen3vl/model.rs:1292-1308); rows `(n, 1536)` f32, grids `(n, 3)`
    Vision { image: Option<(&'a [f32], usize, &'a [u32])>, video: Option<(&'a [f32], usize, &'a [u32])> },
    /// Qwen3-ASR: Whisper log-mel `(num_mel_bins, n_frames)` f32 (qwen3_asr/generate.rs:100-125)
    AudioFeatures { features: &'a [f32], n_frames: usize },
    /// Qwen3-ASR: raw 16 kHz mono samples; the library computes the log-mel features on the GPU
    AudioSamples { samples: &'a [f32] },
}

/// One model on one GPU.  Not thread-safe by contract (the reference takes `&mut self` everywhere and a write lock per request).
pub struct Model {
    ctx: *mut sys::AhaCtx,
    model: *mut sys::AhaModel,
    vocab: usize,
    stop: Vec<u32>,
}
unsafe impl Send for Model {}

impl Model {
    /// `XxxGenerateModel::init(path, ..)` minus tokenizer / chat template: the library parses `config.json` /
    /// `generation_config.json` and mmaps every `*.safetensors` of `dir` (qwen3/generate.rs:22-50, utils/mod.rs:121-137).
    /// `dtype`: the resolved `Option<DType>` of `init` as an `aha_dtype` code; anything but bf16 is refused, loudly.
    pub fn from_dir(dir: &str, device: i32, dtype: Option<i32>) -> Result<Self, Error> {
        let c = CString::new(dir).map_err(|_| Error { code: -1, message: "path contains NUL".into() })?;
        let mut desc = sys::AhaModelDesc::default();
        check(unsafe { sys::aha_hip_config_parse(c.as_ptr(), &mut desc) })?;
        // get_dtype(dtype, cfg_dtype) exactly as XxxGenerateModel::init resolves it (utils/mod.rs:77-115): an explicit request wins,
        // otherwise the checkpoint's own dtype string decides -- and whatever comes out must be a dtype the kernels compute in, so
        // a float16 / float32 checkpoint with `dtype == None` is REFUSED here instead of silently running in bf16
        let mut buf = [0 as c_char; 64];
        check(unsafe { sys::aha_hip_config_torch_dtype(c.as_ptr(), buf.as_mut_ptr(), buf.len()) })?;
        let cfg_dtype = unsafe { CStr::from_ptr(buf.as_ptr()) }.to_string_lossy().into_owned();
        let resolved = get_dtype(dtype, &cfg_dtype)?;
        check(unsafe { sys::aha_hip_check_dtype(resolved) })?;
        let mut ctx = std::ptr::null_mut();
        check(unsafe { sys::aha_hip_init(device, &mut ctx) })?;
        let mut model = std::ptr::null_mut();
        if let Err(e) = check(unsafe { sys::aha_hip_model_load(ctx, c.as_ptr(), 0, &mut model) }) {
            unsafe { sys::aha_hip_shutdown(ctx) };
            return Err(e);
        }
        let mut stop = vec![0u32; 8];
        let n = unsafe { sys::aha_hip_stop_token_ids(model, stop.as_mut_ptr(), 8) };
        stop.truncate(n.clamp(0, 8) as usize);
        Ok(Self { ctx, model, vocab: desc.vocab_size as usize, stop })
    }

    /// From tensors the caller already holds (e.g. the mmapped safetensors views `VarBuilder::from_mmaped_safetensors` opens):
    /// `(HF name, bytes, aha_dtype, shape)`; copied to HBM, never aliased.
    pub fn from_tensors(desc: &sys::AhaModelDesc, tensors: &[(String, &[u8], i32, Vec<usize>)], device: i32) -> Result<Self, Error> {
        let names: Vec<CString> = tensors.iter().map(|t| CString::new(t.0.as_str()).unwrap_or_default()).collect();
        let views: Vec<sys::AhaTensorView> = tensors
            .iter()
            .zip(&names)
            .map(|(t, n)| {
                let mut shape = [0i64; 5];
                for (i, s) in t.3.iter().take(5).enumerate() {
                    shape[i] = *s as i64;
                }
                sys::AhaTensorView { name: n.as_ptr(), data: t.1.as_ptr() as *const _, dtype: t.2, ndim: t.3.len() as i32, shape, on_device: 0 }
            })
            .collect();
        let mut ctx = std::ptr::null_mut();
        check(unsafe { sys::aha_hip_init(device, &mut ctx) })?;
        let mut model = std::ptr::null_mut();
        if let Err(e) = check(unsafe { sys::aha_hip_model_create(ctx, desc, views.as_ptr(), views.len(), &mut model) }) {
            unsafe { sys::aha_hip_shutdown(ctx) };
            return Err(e);
        }
        let n = desc.n_stop_tokens.clamp(0, 8) as usize;
        Ok(Self { ctx, model, vocab: desc.vocab_size as usize, stop: desc.stop_tokens[..n].to_vec() })
    }

    pub fn vocab_size(&self) -> usize {
        self.vocab
    }
    pub fn stop_token_ids(&self) -> Vec<u32> {
        self.stop.clone()
    }
    pub fn cache_len(&self) -> usize {
        unsafe { sys::aha_hip_cache_len(self.model) }
    }

    /// `InferenceModel::forward_initial`: logits of the LAST prompt position (V floats) and their first-max arg-max.
    /// `logits = None` skips the 608 KB device-to-host copy (greedy requests, or the sampled path via `sample_candidates`).
    pub fn forward_initial(&mut self, ids: &[u32], seqlen_offset: usize, mm: MmInput<'_>, logits: Option<&mut [f32]>) -> Result<u32, Error> {
        let mut c = sys::AhaMmInput::empty();
        let mm_ptr: *const sys::AhaMmInput = match mm {
            MmInput::None => std::ptr::null(),
            MmInput::Image { pixel_values, n_patches, grid_thw } => {
                c.pixel_values = pixel_values.as_ptr() as *const _;
                c.pixel_dtype = sys::AHA_F32;
                c.n_patches = n_patches as i64;
                c.image_grid_thw = grid_thw.as_ptr();
                c.n_images = (grid_thw.len() / 3) as i32;
                &c
            }
            MmInput::Vision { image, video } => {
                c.pixel_dtype = sys::AHA_F32;
                if let Some((pv, n, grid)) = image {
                    c.pixel_values = pv.as_ptr() as *const _;
                    c.n_patches = n as i64;
                    c.image_grid_thw = grid.as_ptr();
                    c.n_images = (grid.len() / 3) as i32;
                }
                if let Some((pv, n, grid)) = video {
                    c.pixel_values_video = pv.as_ptr() as *const _;
                    c.n_patches_video = n as i64;
                    c.video_grid_thw = grid.as_ptr();
                    c.n_videos = (grid.len() / 3) as i32;
                }
                &c
            }
            MmInput::AudioFeatures { features, n_frames } => {
                c.audio_features = features.as_ptr();
                c.n_frames = n_frames as i64;
                &c
            }
            MmInput::AudioSamples { samples } => {
                c.audio_samples = samples.as_ptr();
                c.n_samples = samples.len() as i64;
                &c
            }
        };
        let lp = match logits {
            Some(l) => {
                assert!(l.len() >= self.vocab);
                l.as_mut_ptr()
            }
            None => std::ptr::null_mut(),
        };
        let mut am = 0u32;
        check(unsafe { sys::aha_hip_forward_initial(self.model, ids.as_ptr(), ids.len(), seqlen_offset, mm_ptr, lp, &mut am) })?;
        Ok(am)
    }

    /// `InferenceModel::forward_step` for the `(1, 1)` token tensor of a decode step.
    pub fn forward_step(&mut self, token: u32, seqlen_offset: usize, logits: Option<&mut [f32]>) -> Result<u32, Error> {
        let lp = match logits {
            Some(l) => {
                assert!(l.len() >= self.vocab);
                l.as_mut_ptr()
            }
            None => std::ptr::null_mut(),
        };
        let mut am = 0u32;
        check(unsafe { sys::aha_hip_forward_step(self.model, token, seqlen_offset, lp, &mut am) })?;
        Ok(am)
    }

    pub fn clear_cache(&mut self) {
        unsafe { sys::aha_hip_clear_cache(self.model) };
    }

    /// The greedy loop of `generate_generic` / `generate_stream_generic` (common/generate.rs:115-159, 161-368) kept on the
    /// device in chunks of `chunk` tokens: `on_token` sees every token in order (what a streaming response forwards) and
    /// returns `false` to stop; an eos id stops after it has been delivered, as in the reference.
    pub fn decode_greedy_stream<F: FnMut(u32) -> bool>(
        &mut self,
        first_token: u32,
        mut seqlen_offset: usize,
        max_new: usize,
        chunk: usize,
        mut on_token: F,
    ) -> Result<usize, Error> {
        let mut buf = vec![0u32; chunk.max(1)];
        let (mut tok, mut produced) = (first_token, 0usize);
        while produced < max_new {
            let n = buf.len().min(max_new - produced);
            let got = unsafe { sys::aha_hip_decode_greedy(self.model, tok, seqlen_offset, n, buf.as_mut_ptr()) };
            check(got)?;
            let got = got as usize;
            for &t in &buf[..got] {
                produced += 1;
                if !on_token(t) || self.stop.contains(&t) {
                    return Ok(produced);
                }
            }
            if got < n || got == 0 {
                break; // the library saw an eos id inside the chunk
            }
            tok = buf[got - 1];
            seqlen_offset += got;
        }
        Ok(produced)
    }

    /// `sample_and_push` without the logits copy (common/generate.rs:70-86): repeat penalty over `context`, the `k` largest
    /// penalised logits (value descending, index ascending), the full-vocabulary max and sum exp((x - max) / T).
    pub fn sample_candidates(&mut self, context: &[u32], repeat_penalty: f32, temperature: f32, k: usize) -> Result<(Vec<f32>, Vec<u32>, f32, f32), Error> {
        let (mut vals, mut idx) = (vec![0f32; k], vec![0u32; k]);
        let (mut mx, mut se) = (0f32, 0f32);
        check(unsafe {
            sys::aha_hip_sample_candidates(self.model, context.as_ptr(), context.len(), repeat_penalty, temperature, k as i32, vals.as_mut_ptr(), idx.as_mut_ptr(), &mut mx, &mut se)
        })?;
        Ok((vals, idx, mx, se))
    }

    pub fn last_logits(&mut self, out: &mut [f32]) -> Result<(), Error> {
        assert!(out.len() >= self.vocab);
        check(unsafe { sys::aha_hip_last_logits(self.model, out.as_mut_ptr()) })
    }

    /// 128 bytes that identify an RCCL communicator: rank 0 makes them, the host sends them to every rank
    pub fn rccl_unique_id() -> Result<[u8; 128], Error> {
        let mut id = [0u8; 128];
        check(unsafe { sys::aha_hip_tp_unique_id(id.as_mut_ptr() as *mut std::ffi::c_void) })?;
        Ok(id)
    }

    /// Context-parallel prefill over `world` GPUs of one node (one process / one `Model` with the FULL weights per GPU): after this,
    /// every rank calls `forward_initial` with the same prompt at offset 0; each ends with the whole KV cache and the same logits, and
    /// decode continues on rank 0 exactly as after a single-GPU prefill (no hand-back).  `unique_id`: `rccl_unique_id()` of rank 0.
    pub fn set_context_parallel(&mut self, rank: usize, world: usize, unique_id: &[u8; 128]) -> Result<(), Error> {
        check(unsafe { sys::aha_hip_set_context_parallel(self.model, rank as i32, world as i32, None, std::ptr::null_mut()) })?;
        if world > 1 {
            check(unsafe { sys::aha_hip_cp_init_rccl(self.model, unique_id.as_ptr() as *const std::ffi::c_void) })?;
        }
        Ok(())
    }
}

impl Drop for Model {
    fn drop(&mut self) {
        unsafe {
            sys::aha_hip_model_destroy(self.model);
            sys::aha_hip_shutdown(self.ctx);
        }
    }
}

/// candle's `LogitsProcessor::rng` + `sample_multinomial` behind the C ABI (`aha_hip_rng_*`): rand 0.9.2 `StdRng::seed_from_u64`
/// (candle-transformers' own rand, Cargo.lock:590-606) and `WeightedIndex::<f32>::new(prs)?.sample(&mut rng)`.  A host that keeps
/// candle's `LogitsProcessor` does not need this; a host that takes `aha_hip_sample_candidates`' candidates instead of the
/// V-float logits draws with it so that a seed still defines the token sequence.
pub struct StdRng(*mut sys::AhaRng);
unsafe impl Send for StdRng {}
impl StdRng {
    pub fn seed_from_u64(seed: u64) -> Result<Self, Error> {
        let mut h = std::ptr::null_mut();
        check(unsafe { sys::aha_hip_rng_create(seed, &mut h) })?;
        Ok(Self(h))
    }
    pub fn next_u32(&mut self) -> u32 {
        unsafe { sys::aha_hip_rng_next_u32(self.0) }
    }
    /// `WeightedIndex::new(weights)?.sample(self)`: Err for a negative / NaN weight or an all-zero vector, the stream untouched.
    pub fn weighted_index(&mut self, weights: &[f32]) -> Result<usize, Error> {
        let mut idx = 0u32;
        check(unsafe { sys::aha_hip_rng_weighted_index(self.0, weights.as_ptr(), weights.len(), &mut idx) })?;
        Ok(idx as usize)
    }
}
impl Drop for StdRng {
    fn drop(&mut self) {
        unsafe { sys::aha_hip_rng_destroy(self.0) }
    }
}

/// The reference's seam: `trait InferenceModel` (src/models/common/mod.rs:25-45).
#[cfg(feature = "aha")]
pub mod inference_model {
    use super::{MmInput, Model};
    use aha::models::common::{InferenceModel, MultiModalData};
    use anyhow::{anyhow, Result};
    use candle_core::{DType, Device, Tensor};

    pub struct HipQwen3 {
        pub model: Model,
    }

    impl HipQwen3 {
        fn logits_tensor(&self, v: Vec<f32>) -> Result<Tensor> {
            // generate.rs:75 squeezes the (1, 1, V) tensor and casts it to f32 on the host: hand it over that way
            Ok(Tensor::from_vec(v, (1, 1, self.model.vocab_size()), &Device::Cpu)?)
        }
    }

    impl InferenceModel for HipQwen3 {
        fn forward_initial(&mut self, input_ids: &Tensor, seqlen_offset: usize, data: MultiModalData) -> Result<Tensor> {
            let ids = input_ids.flatten_all()?.to_vec1::<u32>()?;
            let mut logits = vec![0f32; self.model.vocab_size()];
            let first = data.data_vec.first().cloned().flatten();
            let second = data.data_vec.get(1).cloned().flatten();
            let third = data.data_vec.get(2).cloned().flatten();
            let fourth = data.data_vec.get(3).cloned().flatten();
            let (host_a, host_b, host_c, host_d);
            let mm = match (first, second) {
                // Qwen3-VL with a video: data_vec = [pixel_values?, image_grid_thw?, pixel_values_video, video_grid_thw,
                // cache_position] (qwen3vl/model.rs:1292-1308)
                (img, igrid) if third.is_some() && fourth.is_some() => {
                    let (pvv, vgrid) = (third.unwrap(), fourth.unwrap());
                    host_c = pvv.to_dtype(DType::F32)?.flatten_all()?.to_vec1::<f32>()?;
                    host_d = vgrid.flatten_all()?.to_vec1::<u32>()?;
                    let image = match (img, igrid) {
                        (Some(pv), Some(grid)) => {
                            host_a = pv.to_dtype(DType::F32)?.flatten_all()?.to_vec1::<f32>()?;
                            host_b = grid.flatten_all()?.to_vec1::<u32>()?;
                            Some((&host_a[..], pv.dim(0)?, &host_b[..]))
                        }
                        _ => None,
                    };
                    MmInput::Vision { image, video: Some((&host_c[..], pvv.dim(0)?, &host_d[..])) }
                }
                // Qwen3-VL: data_vec = [pixel_values, image_grid_thw, None, None, cache_position] (qwen3vl/generate.rs:79-101)
                (Some(pv), Some(grid)) => {
                    host_a = pv.to_dtype(DType::F32)?.flatten_all()?.to_vec1::<f32>()?;
                    host_b = grid.flatten_all()?.to_vec1::<u32>()?;
                    MmInput::Image { pixel_values: &host_a, n_patches: pv.dim(0)?, grid_thw: &host_b }
                }
                // Qwen3-ASR: data_vec = [input_features] (num_mel_bins, n_frames) (qwen3_asr/generate.rs:100-125)
                (Some(feat), None) => {
                    let n_frames = feat.dim(feat.rank() - 1)?;
                    host_a = feat.to_dtype(DType::F32)?.flatten_all()?.to_vec1::<f32>()?;
                    MmInput::AudioFeatures { features: &host_a, n_frames }
                }
                _ => MmInput::None,
            };
            self.model.forward_initial(&ids, seqlen_offset, mm, Some(&mut logits)).map_err(|e| anyhow!(e.to_string()))?;
            self.logits_tensor(logits)
        }

        fn forward_step(&mut self, input_ids: &Tensor, seqlen_offset: usize) -> Result<Tensor> {
            let tok = input_ids.flatten_all()?.to_vec1::<u32>()?[0];
            let mut logits = vec![0f32; self.model.vocab_size()];
            self.model.forward_step(tok, seqlen_offset, Some(&mut logits)).map_err(|e| anyhow!(e.to_string()))?;
            self.logits_tensor(logits)
        }

        fn clear_cache(&mut self) {
            self.model.clear_cache();
        }

        fn stop_token_ids(&self) -> Vec<u32> {
            self.model.stop_token_ids()
        }
    }
}
