/*
 * metalchat_hip.h -- C ABI of the MI355X (gfx950) backend for metalchat's decode hot path.
 *
 * This is the drop-in boundary: plain C, opaque handles, pointers and sizes only.  It exports
 * exactly what the reference's Metal seam binds for this path, one entry point per Metal call
 * site, plus the fused decode pipeline that sits behind the same handles.  Reference citations
 * are paths relative to the metalchat source tree (v1.2.1).
 *
 * Part 1 mirrors  metal::{device,library,kernel,buffer}   include/metalchat/metal.h:14-34
 *                 src/metal.cc:21-84, src/metal_impl.h:19-120
 *                 hardware_function_encoder / kernel_thread  include/metalchat/kernel_thread.h:57-294
 *                 src/kernel_thread.cc:13-223
 * Part 2 is the decode pipeline driven by nn::llama3 / nn::gemma3 / transformer<Layer>::transform
 *                 include/metalchat/nn/llama.h:113-134, nn/gemma.h:110-137, transformer.h:357-364
 *
 * Error convention: every function that can fail returns an mc_status (0 = ok).  The message of
 * the last failure on the calling thread is mc_last_error().  The C++ shim rethrows
 * MC_ERR_INVALID_ARGUMENT as std::invalid_argument, MC_ERR_RUNTIME as std::runtime_error and
 * MC_ERR_ALLOC as alloc_error with the same message text the reference uses
 * (include/metalchat/tensor/expected.h:185-193, include/metalchat/allocator.h:20-34).
 *
 * There is NO CPU fallback anywhere behind this ABI: without a HIP device every entry point that
 * needs one fails with MC_ERR_RUNTIME.
 */
#ifndef METALCHAT_HIP_H
#define METALCHAT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t mc_status;
enum {
    MC_OK = 0,
    MC_ERR_INVALID_ARGUMENT = 1, /* std::invalid_argument */
    MC_ERR_RUNTIME = 2,          /* std::runtime_error    */
    MC_ERR_ALLOC = 3             /* alloc_error           */
};

const char* mc_last_error(void);
/* "metalchat-hip <version> gfx950" */
const char* mc_version(void);
/* 1 when every kernel launch is wrapped in a named range "name<grid,group>" (roctxRangePush / Pop; MC_TRACE_RANGES=1 and
 * libroctx64 found) -- the reference labels every encoder that way for GPU capture, src/kernel_thread.cc:109-115 */
int32_t mc_trace_ranges_enabled(void);

/* ------------------------------------------------------------------------------------------
 * Part 1 -- backend seam
 * ------------------------------------------------------------------------------------------ */
typedef struct mc_device mc_device;
typedef struct mc_library mc_library;
typedef struct mc_kernel mc_kernel;
typedef struct mc_buffer mc_buffer;
typedef struct mc_queue mc_queue;

/* MTL::CreateSystemDefaultDevice -- src/metal.cc:51-55.  ordinal < 0 selects the current HIP
 * device (LOCAL_RANK-aware callers pass their own ordinal). */
mc_status mc_device_create(int32_t ordinal, mc_device** out);
void mc_device_release(mc_device* dev);
/* HIP devices visible to this process (the reference has exactly one: MTL::CreateSystemDefaultDevice, src/metal.cc:51-55);
 * the layer pipeline puts one stage on each. */
int32_t mc_device_count(void);
/* device->name() -- src/accelerator.cc:108-113 */
const char* mc_device_name(const mc_device* dev);
/* device->maxBufferLength() -- src/accelerator.cc:73-77 */
size_t mc_device_max_buffer_size(const mc_device* dev);
int32_t mc_device_ordinal(const mc_device* dev);
int32_t mc_device_compute_units(const mc_device* dev);

/* device->newLibrary(url) -- src/metal.cc:58-84.  `path` is a gfx950 code object (.hsaco).
 * Failure message is "metal: library not found"-style: "hip: library not found" when the file
 * is absent (test/test_accelerator.cc:15-21 expects the runtime_error). */
mc_status mc_library_open(mc_device* dev, const char* path, mc_library** out);
void mc_library_release(mc_library* lib);

/* library->newFunction(name) + newComputePipelineState -- src/accelerator.cc:116-158.
 * Name format {kernel}[_{block}]_{type}[_{type}...] with types bfloat,float,int32_t,int8_t
 * (include/metalchat/accelerator.h:175-218, include/metalchat/dtype.h:83-118).
 * Missing function -> MC_ERR_INVALID_ARGUMENT "hardware_accelerator: function <name> not found
 * in a shader library". */
mc_status mc_library_get_kernel(mc_library* lib, const char* name, mc_kernel** out);
void mc_kernel_release(mc_kernel* k);
const char* mc_kernel_name(const mc_kernel* k);
/* pipeline->maxTotalThreadsPerThreadgroup() -- src/kernel.cc:75-79 */
size_t mc_kernel_max_threads_per_group(const mc_kernel* k);

/* device->newBuffer(bytes, Shared) / newBuffer(ptr, bytes, Shared) / newBuffer(ptr, bytes,
 * Shared, nil) (no-copy) -- src/allocator.cc:127-173.  HBM-resident (hipMalloc); the no-copy
 * form wraps an existing DEVICE pointer (for example torch-owned memory) without owning it. */
mc_status mc_buffer_alloc(mc_device* dev, size_t bytes, mc_buffer** out);
mc_status mc_buffer_alloc_copy(mc_device* dev, const void* host_src, size_t bytes, mc_buffer** out);
mc_status mc_buffer_wrap_nocopy(mc_device* dev, void* device_ptr, size_t bytes, mc_buffer** out);
void mc_buffer_release(mc_buffer* buf);
/* buffer->contents() / buffer->length() -- src/metal.cc:21-32.  contents() is a DEVICE address:
 * weights live in HBM, so the reference's "CPU dereferences the shared buffer" idiom becomes the
 * explicit copies below ("Unified memory assumption"). */
void* mc_buffer_contents(const mc_buffer* buf);
size_t mc_buffer_length(const mc_buffer* buf);
mc_status mc_buffer_upload(mc_buffer* buf, size_t offset, const void* host_src, size_t bytes);
mc_status mc_buffer_download(const mc_buffer* buf, size_t offset, void* host_dst, size_t bytes);
mc_status mc_buffer_fill_zero(mc_buffer* buf, size_t offset, size_t bytes);

/* device->newCommandQueue() + commandBuffer + computeCommandEncoder -- src/kernel_thread.cc:23-32.
 * One in-order HIP stream subsumes the MTL::Event chain of src/kernel_thread.cc:34-47,194 and
 * the per-argument memoryBarrier of include/metalchat/kernel_thread.h:121-124.
 * external_stream != NULL adopts a caller-owned hipStream_t (for example torch's current stream). */
mc_status mc_queue_create(mc_device* dev, void* external_stream, mc_queue** out);
void mc_queue_release(mc_queue* q);
void* mc_queue_stream(const mc_queue* q);

/* encoder->setComputePipelineState -- src/kernel_thread.cc:69-74.  Resets the argument index. */
mc_status mc_encoder_set_kernel(mc_queue* q, mc_kernel* k);
/* encoder->setBytes(data, size, index++) -- src/kernel_thread.cc:77-81.  The argument is placed at
 * the next offset aligned to min(size, 4) rounded down to a power of two (tensor_layout<N> and
 * 32-bit scalars: 4; a bfloat scalar: 2). */
mc_status mc_encoder_set_bytes(mc_queue* q, const void* data, size_t size);
/* encoder->setBuffer(buffer, offset, index++) -- src/kernel_thread.cc:84-88.  offset in BYTES. */
mc_status mc_encoder_set_buffer(mc_queue* q, mc_buffer* buf, size_t byte_offset);
/* encoder->memoryBarrier(resource) -- src/kernel_thread.cc:91-96.  A no-op on an in-order stream;
 * kept so the reference's encode() sequence maps 1:1. */
mc_status mc_encoder_memory_barrier(mc_queue* q, mc_buffer* buf);
/* encoder->dispatchThreads(grid, group) -- src/kernel_thread.cc:105-120.  grid = TOTAL threads,
 * group = threads per group; blocks = ceil(grid/group) per axis and kernels bounds-check, which is
 * how Metal's non-uniform threadgroups behave.  Validation as include/metalchat/kernel.h:119-141:
 * group.numel() <= max_threads and grid.numel() >= group.numel() else MC_ERR_INVALID_ARGUMENT. */
mc_status mc_encoder_dispatch_threads(mc_queue* q, const size_t grid[3], const size_t group[3]);
/* Same, with an explicit dynamic LDS size (fused kernels only). */
mc_status mc_encoder_dispatch_threads_lds(mc_queue* q, const size_t grid[3], const size_t group[3],
                                          size_t lds_bytes);
/* commandBuffer->addCompletedHandler -- src/kernel_thread.cc:134-144.  The callback runs on a
 * driver thread once everything submitted before it has completed; status != MC_OK carries a GPU
 * execution error (the reference delivers it through the promise). */
typedef void (*mc_completion_fn)(void* ctx, mc_status status);
mc_status mc_queue_on_completed(mc_queue* q, mc_completion_fn fn, void* ctx);
/* commandBuffer->commit() -- src/kernel_thread.cc:184-199.  Launches are asynchronous already;
 * commit only flushes.  mc_queue_wait is future_tensor::wait (include/metalchat/tensor/future.h:235-253). */
mc_status mc_queue_commit(mc_queue* q);
mc_status mc_queue_wait(mc_queue* q);

/* Timing of a region of the queue with HIP events recorded ON THE QUEUE'S STREAM (bench.py).
 * begin/end record the events, elapsed synchronises on the end event. */
mc_status mc_queue_timer_begin(mc_queue* q);
mc_status mc_queue_timer_end(mc_queue* q);
mc_status mc_queue_timer_elapsed_ms(mc_queue* q, float* ms);

/* ------------------------------------------------------------------------------------------
 * Part 2 -- fused decode pipeline (what nn::llama3::operator() does per token, behind the seam)
 * ------------------------------------------------------------------------------------------ */
enum { MC_DTYPE_BF16 = 0, MC_DTYPE_F32 = 1 };
enum { MC_FAMILY_LLAMA3 = 0, MC_FAMILY_GEMMA3 = 1 };
/* weight formats of a linear layer held in HBM */
enum {
    MC_WFMT_T = 0,   /* nn::linear: row-major [out,in] of T                                    */
    MC_WFMT_I8 = 1,  /* int8 [out,in] + per-(row,group) scale  (quantization::lora_linear)      */
    MC_WFMT_I4 = 2   /* packed int4 (two values per byte, offset-binary) + per-(row,group) scale */
};
/* quantised-GEMV arithmetic */
enum {
    MC_QMODE_EXACT = 0, /* Wd = T(T(q)*T(s)) materialised per weight exactly as
                           hadamard_broadcast does (kernel/mul.metal:78-82), then fp32 dot        */
    MC_QMODE_FAST = 1   /* y = sum_g s_g * sum_k q_k x_k : skips the per-weight rounding to T     */
};

typedef struct {
    int32_t dtype;   /* activation / KV / norm-weight type T */
    int32_t family;
    int32_t dim, n_heads, n_kv_heads, head_dim, ffn_dim, n_layers, vocab, max_seq_len;
    float rope_theta;
    float rope_sliding_theta; /* gemma3: theta of the sliding layers, 0 otherwise */
    int32_t sliding_stride;   /* gemma3: layer i is sliding iff (i+1) % stride != 0 */
    float norm_eps;
    float attn_scale;
    int32_t sink_pre_len;     /* < 0: bit_width(max_seq_len) - 1 (include/metalchat/nn/cache.h:125-127) */
    /* pipeline partition: this decoder owns layers [layer_begin, layer_end); the first stage
     * owns the embedding, the last stage owns final norm + output head. */
    int32_t layer_begin, layer_end;
    int32_t weight_format;    /* MC_WFMT_* of the seven per-layer linears and the output head */
    int32_t group_size;       /* quantisation group along `in` (per-row scale: group_size = 0)   */
    int32_t qmode;            /* MC_QMODE_* */
    int32_t use_graph;        /* capture one token step into a hipGraph and replay it */
} mc_decoder_config;

typedef struct mc_decoder mc_decoder;

mc_status mc_decoder_create(mc_device* dev, mc_library* lib, mc_queue* q,
                            const mc_decoder_config* cfg, mc_decoder** out);
void mc_decoder_release(mc_decoder* d);

/* Names of the per-layer tensors: "wq","wk","wv","wo","w1","w2","w3" (linears),
 * "attention_norm","ffn_norm","q_norm","k_norm","attention_post_norm","ffn_post_norm" (T[n]).
 * layer = -1 addresses the model-level tensors "tok_embeddings", "output" (linears) and "norm".
 *
 * mc_decoder_load_linear takes the REFERENCE-NATIVE host format and repacks it for HBM:
 *   MC_WFMT_T : weight = T[out*in]
 *   MC_WFMT_I8/I4 : weight = int8_t[out*in] (int4 range [-8,7] held one value per byte, as the
 *                   reference loads it: include/metalchat/huggingface/llama.h:159-168),
 *                   scales = float[out * in/group] (group 0: float[out])
 * For "tok_embeddings" a quantised table is always kept as int8 + per-row scale
 * (quantization::lora_embedding, include/metalchat/quantization/lora.h:133-175). */
mc_status mc_decoder_load_linear(mc_decoder* d, int32_t layer, const char* name,
                                 int32_t weight_format, int32_t out_features, int32_t in_features,
                                 int32_t group_size, const void* weight, const float* scales);
/* quantization::lora_adaptor of one projection (include/metalchat/quantization/lora.h:17-53,
 * 119-121): result = T(T(x Wd^T) + T(T(B(A x)) * scale)).  name as in mc_decoder_load_linear
 * ("wq","wk","wv","wo","w1","w3","w2"); a_T = T[rank * in_features], b_T = T[out_features * rank]
 * (nn::linear weights, row-major [out, in]).  Adaptors of projections that share a fused GEMV
 * (wq|wk|wv, w1|w3) must share rank and scale; rank is a multiple of 8. */
mc_status mc_decoder_load_lora(mc_decoder* d, int32_t layer, const char* name, int32_t rank,
                               int32_t out_features, int32_t in_features, const void* a_T,
                               const void* b_T, float scale);
mc_status mc_decoder_load_vector(mc_decoder* d, int32_t layer, const char* name, int32_t n,
                                 const void* data_T);
/* Synthetic weights generated ON THE DEVICE from a counter-based hash (bench.py at full model
 * sizes; tests regenerate any row on the host with mc_synth_* below and check it). */
mc_status mc_decoder_init_synthetic(mc_decoder* d, uint64_t seed);

/* transformer<Layer>::transform(token, start_pos) with len == 1 -- include/metalchat/transformer.h:357-364.
 * Runs the layers this decoder owns.  First stage: `token` is embedded.  Other stages: the hidden
 * row is read from hidden_in (device pointer, T[dim]).  Last stage: writes the greedy next token
 * to *next_token (after a stream sync) when next_token != NULL.  Non-last stages leave the hidden
 * row in the buffer returned by mc_decoder_hidden_out(). */
mc_status mc_decoder_step(mc_decoder* d, int32_t token, int32_t start_pos, const void* hidden_in,
                          int32_t* next_token);
/* The prompt pass: transformer<Layer>::transform(input[1, len], start_pos) with len > 1 --
 * nn::llama3 / nn::gemma3 operator() (include/metalchat/nn/llama.h:113-134, nn/gemma.h:110-137).
 * All len rows go through the layers as GEMMs (weights dequantised once per tile), the K / V rows
 * are written to cache positions [start_pos, start_pos + len), scores take the reference's mask
 * (make_causal_mask / make_sliding_causal_mask, nn/attention.h:283-321: only the LAST len columns
 * form the causal square, columns of an earlier context stay masked -- as the reference builds it),
 * and the last row goes through the head and the sampler; *next_token receives its pick.
 * sliding_window: gemma3_options.sliding_window (sliding layers only); 0 for llama3.
 * len == 1 is mc_decoder_step.  The cache follows nn::sink_cache::copy (nn/cache.h:167-216): a chunk either fits
 * behind start_pos (start_pos + len <= max_seq_len), or, once start_pos >= max_seq_len, the post-sink region is
 * rotated left by len and the chunk takes the last len rows (len <= max_seq_len, "sink_cache: requested length ... is
 * larger than the cache size" otherwise).  A chunk that starts inside the cache and ends outside is an error, as it
 * is in the reference (its cache slice runs out of range).
 * Long prompts: a GEMM launch with >= 48 tiles of 256 x 256 (MC_PF_BLASLT_TILES) multiplies a dequantised bfloat16 copy of the
 * matrix -- Wd = T(T(q) T(s)), kernel/mul.metal:78-82, 2 bytes per weight more device memory, built on first use -- in the ROCm
 * library GEMM (libhipblaslt, opened with dlopen on first use; MC_PF_BLASLT=0 or an absent library: the prompt kernels alone).
 * Same operand values, fp32 sums rounded to T once (nn/linear.h:70-81); the first such prompt of a process also pays the library's start-up.
 * mc_decoder_prefill_stage is the same pass on ONE STAGE of a layer pipeline: the first stage takes `tokens`, a later
 * stage the [len][dim] hidden rows of the previous one (`rows_in`, device memory); *rows_out (device memory, valid
 * until the next prompt pass) receives this stage's rows; the last stage runs the head and fills *next_token. */
mc_status mc_decoder_prefill(mc_decoder* d, const int32_t* tokens, int32_t len, int32_t start_pos,
                             int32_t sliding_window, int32_t* next_token);
mc_status mc_decoder_prefill_stage(mc_decoder* d, const int32_t* tokens, const void* rows_in, int32_t len,
                                   int32_t start_pos, int32_t sliding_window, void** rows_out, int32_t* next_token);
/* Enqueue `n` chained greedy steps entirely on the device (token feedback through HBM, one host
 * sync at the end); tokens_out receives the n generated ids.  Single-stage decoders only. */
mc_status mc_decoder_generate(mc_decoder* d, int32_t first_token, int32_t start_pos, int32_t n,
                              int32_t* tokens_out);
void* mc_decoder_hidden_out(mc_decoder* d);
void* mc_decoder_hidden_in(mc_decoder* d);

/* Layer pipeline over N stages: stage r owns a contiguous range of L / N layers (the first L % N
 * stages one more: mc_pipeline_layer_range) of the strict layer
 * chain include/metalchat/nn/llama.h:123-126 and their caches (nn/attention.h:122-130); per token the hidden row
 * hops stage to stage and the 4-byte pick returns to stage 0.  Everything, hops included, is enqueued on the
 * stages' streams: one host synchronisation per mc_pipeline_generate call.
 *   mc_pipeline_unique_id / _create : one process per GPU; the hop is ONE ncclSend / ncclRecv pair (RCCL over
 *       xGMI) on the decoder's own stream.  Rank 0 makes the 128-byte id, every rank passes the same bytes.
 *   mc_pipeline_create_local        : all N stages in this process (on one device or several): the hop is a
 *       device-to-device copy behind an event -- the same launches per stage, runnable on a one-GPU box.
 *   mc_pipeline_generate            : called by EVERY rank with the same arguments; tokens_out is filled on
 *       rank 0 and on the last rank (local: always).  Greedy or the last stage's sampler.
 *   mc_pipeline_allreduce_max       : barrier + device synchronise, then *value = max over ranks. */
typedef struct mc_pipeline mc_pipeline;
void mc_pipeline_layer_range(int32_t rank, int32_t world, int32_t n_layers, int32_t* layer_begin, int32_t* layer_end);
mc_status mc_pipeline_unique_id(void* id_128_bytes);
mc_status mc_pipeline_create(mc_decoder* stage, int32_t rank, int32_t world, const void* id_128_bytes, mc_pipeline** out);
mc_status mc_pipeline_create_local(mc_decoder** stages, int32_t n, mc_pipeline** out);
mc_status mc_pipeline_generate(mc_pipeline* p, int32_t first_token, int32_t start_pos, int32_t n, int32_t* tokens_out);
mc_status mc_pipeline_prefill(mc_pipeline* p, const int32_t* tokens, int32_t len, int32_t start_pos, int32_t sliding_window,
                              int32_t* next_token); /* the prompt pass: [len][dim] rows hop stage to stage */
mc_status mc_pipeline_allreduce_max(mc_pipeline* p, double* value);
/* What the transport itself reports: the communicator's size and this process's rank in it (ncclCommCount /
 * ncclCommUserRank), so a caller can verify that RCCL saw N ranks; (-1, -1) for a local pipeline. */
mc_status mc_pipeline_comm_info(mc_pipeline* p, int32_t* ranks, int32_t* rank);
void mc_pipeline_release(mc_pipeline* p);
/* Sampler of the last stage -- include/metalchat/nn/sampling.h:152-315.
 *   MC_SAMPLER_GREEDY : argmax, first maximum (the BASELINE configuration).
 *   MC_SAMPLER_DEFAULT: make_default_sampler() = topk_sampler(top_k) -> nucleus_sampler(temperature,
 *                       top_p) -> multinomial_sampler(1), run on the device in two launches with no
 *                       host synchronisation (the reference synchronises three times per token and
 *                       partial_sorts the vocabulary on the CPU).  Errors as nucleus_sampler's
 *                       constructor ("temperature must be positive", "probability must be in
 *                       [0.0, 1.0]").  top_k <= 128.  Equal logits are ordered by index
 *                       (std::partial_sort leaves them unspecified).
 * mc_decoder_set_seeds: the (init_state, init_seq) pairs kernel::multinomial draws from its
 * std::mt19937 per call (include/metalchat/kernel/multinomial.h:46-55); token i of one
 * mc_decoder_generate call uses pair i % n_pairs (mc_decoder_step: pair 0); no pairs = (0, 0). */
enum { MC_SAMPLER_GREEDY = 0, MC_SAMPLER_DEFAULT = 1 };
mc_status mc_decoder_set_sampler(mc_decoder* d, int32_t kind, int32_t top_k, float temperature, float top_p);
mc_status mc_decoder_set_seeds(mc_decoder* d, const uint64_t* seeds, int32_t n_pairs);
/* After a step with taps enabled: [7][k] floats -- scaled logits, probabilities, sorted, cumsum,
 * cumsum - sorted, masked, vocabulary ids -- of the sampler chain. */
mc_status mc_decoder_get_sampler_taps(mc_decoder* d, float* out_7xk);
/* Debug / parity taps (synchronise the queue).  mc_decoder_set_taps(1) makes every step also
 * copy the hidden row after the embedding and after each owned layer into a tap buffer. */
mc_status mc_decoder_set_taps(mc_decoder* d, int32_t enable);
mc_status mc_decoder_get_logits(mc_decoder* d, void* logits_T_vocab);
mc_status mc_decoder_get_hidden(mc_decoder* d, int32_t layer, void* hidden_T_dim);
/* Logical sink-cache view of `layer` after the last step, in the reference's layout
 * [end_pos, n_kv_heads, head_dim] (include/metalchat/nn/cache.h:209-215); *n_valid = end_pos. */
mc_status mc_decoder_export_kv(mc_decoder* d, int32_t layer, void* keys, void* values,
                               int32_t* n_valid);
/* Test aid, the inverse of mc_decoder_export_kv: fill `layer`'s cache with n_valid logical rows
 * [n_valid, n_kv_heads, head_dim] of T as if positions 0 .. n_valid-1 had been decoded (the state
 * sink_cache::copy's first branch leaves, include/metalchat/nn/cache.h:206-213; n_valid <=
 * max_seq_len, the ring is reset).  A step at start_pos = n_valid continues from there, so a parity
 * test can start at the benchmark's context length.  Every owned layer must be imported with the
 * same n_valid. */
mc_status mc_decoder_import_kv(mc_decoder* d, int32_t layer, const void* keys, const void* values,
                               int32_t n_valid);
/* Bytes of weights + scales this decoder streams per token (the roofline numerator). */
size_t mc_decoder_weight_bytes(const mc_decoder* d);
/* Per-kernel timing pass used by bench.py's roofline leg: runs the named fused GEMV of every
 * owned layer back to back between two events on the queue's stream and returns the total
 * milliseconds and the algorithmic bytes moved.  which: "qkv","wo","w13","w2","head","all". */
mc_status mc_decoder_time_gemv(mc_decoder* d, const char* which, int32_t repeats, float* total_ms,
                               double* bytes_per_pass, int32_t* launches_per_pass);
/* Host name of the kernel mc_decoder_time_gemv(which) launches -- the variant a token really runs (linear-order
 * kernels, prologue / epilogue codes, the greedy pick inside the head).  which = "attn": the decode attention kernel(s) of
 * the first owned block ("mc_attn_wo_i4_*" carries the Wo GEMV, "mc_attn_qkv_wo_i4_*" / "mc_attn_qkv_wo_w_*" (plain bfloat weights) the rmsnorm +
 * wq|wk|wv GEMV + RoPE + cache write in front of it as well: the token then launches no Wo (no QKV) GEMV of its own and time_gemv("wo") /
 * time_gemv("qkv") measure stand-alone launches).  Launches nothing. */
mc_status mc_decoder_gemv_kernel_name(mc_decoder* d, const char* which, char* buf, size_t cap);
/* How often an in-launch hand-off of the decode attention gave up (its workgroups were not resident together: another stream or
 * process held part of the chip) and the decoder fell back to the launches that need no co-residency.  A step
 * (mc_decoder_step with a token read back) and a chain that stays inside the cache (mc_decoder_generate, start_pos + n <=
 * max_seq_len) are repeated on those launches and succeed; anywhere else the call that finds the flag fails with
 * MC_ERR_RUNTIME ("... repeat the call") -- the reference delivers a GPU execution error the same way, through the future that
 * is waited for (src/kernel_thread.cc:134-144). */
int32_t mc_decoder_handoff_fallbacks(const mc_decoder* d);
/* The fall-back is temporary: after 256 tokens decoded without a hand-off launch (MC_HANDOFF_REARM=<tokens>, 0 = never) the decoder takes
 * the one-launch blocks again; every further fall-back doubles that distance, so a chip that is shared for good settles on the launches
 * that need no co-residency.  mc_decoder_handoff_rearms: how often that happened; mc_decoder_handoffs_active: 1 while the decoder uses the
 * hand-off launches, 0 while it is degraded (a caller that wants to know why its tokens got slower asks here). */
int32_t mc_decoder_handoff_rearms(const mc_decoder* d);
int32_t mc_decoder_handoffs_active(const mc_decoder* d);
/* HBM bytes this decoder holds in DERIVED copies of its weights, built on demand by the prompt pass: the quad-interleaved int4 copy
 * short prompts stream from (+ 0.5 byte per weight) and the dequantised bfloat16 copy Wd = T(T(q) T(s)) that prompts of 257 rows and
 * more multiply by (+ 2 bytes per weight; built per matrix on first use where the copies of the decoder's blocks fit an eighth of the
 * device's memory and half of what is free -- MC_PF_PLAIN_COPY=1 / 0 forces / forbids; without it the same GEMM dequantises inside its
 * loop, bit for bit the same rows).  0 until a prompt has asked for one.  The reference materialises the dequantised matrix on every
 * call (quantization/lora.h:115-117); here it is built once and rebuilt when the weights change. */
size_t mc_decoder_derived_weight_bytes(const mc_decoder* d);
/* Test aid: record the host names of every kernel the decoder launches from now on (enable = 1 clears the log and
 * drops a captured token graph, whose replay would launch without passing here; 0 stops recording).
 * mc_decoder_launch_log_read copies the newline-separated names and returns the bytes needed (terminator included).
 * The reference labels every encoder with its kernel name for the same purpose (src/kernel_thread.cc:109-115). */
mc_status mc_decoder_launch_log(mc_decoder* d, int32_t enable);
size_t mc_decoder_launch_log_read(mc_decoder* d, char* buf, size_t cap);

/* Device addresses of a fused weight matrix as it lies in HBM: per layer "qkv" (wq|wk|wv rows),
 * "wo", "w13" (w1/w3 rows interleaved), "w2"; layer -1: "output".  For kernel-level tests that
 * drive mc_gemv_* through the encoder (wrap with mc_buffer_wrap_nocopy). */
mc_status mc_decoder_weight_ptrs(mc_decoder* d, int32_t layer, const char* name, void** w,
                                 void** scales, int32_t* rows, int32_t* in_features,
                                 int32_t* ngroups);

/* Host-side helpers shared by tests and the synthetic initialiser. */
/* value in [-7,7] (bits = 4) or [-127,127] (bits = 8), zero mean, of element (row, col) of matrix `matrix_id` */
int32_t mc_synth_weight(uint64_t seed, uint32_t matrix_id, uint32_t row, uint32_t col, int32_t bits);
/* scale of (row, group) : U(0.5,1.5) / (sqrt(in) * 2^(bits-1)) as float */
float mc_synth_scale(uint64_t seed, uint32_t matrix_id, uint32_t row, uint32_t group,
                     int32_t in_features, int32_t bits);
/* T-typed synthetic values before rounding to T.  kind 0: U(0.5,1.5) (norm weights);
 * kind 1: ~N(0,1)*0.02 (embedding rows); kind 2: U(-1,1)/sqrt(n) (plain T linear weights, n = in) */
float mc_synth_value(uint64_t seed, uint32_t matrix_id, uint32_t index, int32_t kind, uint32_t n);
/* Matrix ids of mc_decoder_init_synthetic: layer*16 + {0 wq, 1 wk, 2 wv, 3 wo, 4 w1, 5 w2, 6 w3,
 * 8 attention_norm, 9 ffn_norm, 10 q_norm, 11 k_norm, 12 attention_post_norm, 13 ffn_post_norm};
 * 0xFFFF0000 + {0 tok_embeddings, 1 output, 2 norm}. */

/* ================================================================================================
 * Part 3 -- model files: the data format on the caller side of the decode path.
 *
 *   safetensor_document / sharded_safetensor_document
 *                                   include/metalchat/safetensor.h:534-1030, src/safetensor.cc
 *   checkpoint adaptors             include/metalchat/huggingface/llama.h:85-171,
 *                                   include/metalchat/huggingface/gemma.h:56-84,
 *                                   include/metalchat/reference.h:35-90
 *   option serializers              src/llama.cc:41-55, src/reference.cc:52-66, src/gemma.cc:20-42
 *
 * A document is an ordered list of named tensors over memory-mapped files (host memory; nothing
 * here touches the GPU until mc_decoder_load_document).  Errors follow the reference: a corrupt or
 * unreadable file is MC_ERR_RUNTIME ("safetensor_document: header is corrupted, ..."), a missing
 * name MC_ERR_INVALID_ARGUMENT.
 * ============================================================================================== */
typedef struct mc_document mc_document;

typedef struct mc_tensor_info {
    const char* name;   /* owned by the document */
    const char* dtype;  /* safetensors spelling: "BF16", "F32", "I8", "I32", ... */
    int32_t ndim;
    int64_t shape[8];
    const void* data;   /* host pointer into the mapped file (or the document's own copy) */
    size_t nbytes;
} mc_tensor_info;

enum {
    MC_CKPT_META_LLAMA3 = 0,       /* reference::llama3_safetensor_serializer (Meta names, wq/wk permuted) */
    MC_CKPT_HF_LLAMA3 = 1,         /* huggingface::llama3_safetensor_serializer (model.layers.N.self_attn...) */
    MC_CKPT_META_LLAMA3_QLORA = 2, /* huggingface::llama3_qlora_safetensor_serializer (int8-held int4 + adaptors) */
    MC_CKPT_HF_GEMMA3 = 3          /* huggingface::gemma3_safetensor_serializer */
};

mc_status mc_document_create(mc_document** out);                         /* safetensor_document() */
mc_status mc_document_open(const char* path, mc_document** out);         /* ::open(path) -- src/safetensor.cc:118-136 */
mc_status mc_document_open_sharded(const char* index_json_path, mc_document** out); /* sharded_...::open */
void mc_document_release(mc_document* d);
int32_t mc_document_size(const mc_document* d);                          /* std::distance(begin(), end()) */
/* Entries in document order (a file's tensors by ascending data offset, src/safetensor.cc:111-115). */
mc_status mc_document_tensor(const mc_document* d, int32_t index, mc_tensor_info* out);
mc_status mc_document_find(const mc_document* d, const char* name, mc_tensor_info* out);
/* insert(name, tensor): the bytes are copied.  src/safetensor.cc:182-200 */
mc_status mc_document_insert(mc_document* d, const char* name, const char* dtype, int32_t ndim,
                             const int64_t* shape, const void* data);
/* insert(name, source): a second name for the same storage.  src/safetensor.cc:203-212 */
mc_status mc_document_link(mc_document* d, const char* name, const char* source);
mc_status mc_document_set_metadata(mc_document* d, const char* key, const char* value);
const char* mc_document_metadata(const mc_document* d, const char* key); /* NULL when absent */
/* serializer.adapt(document): rename to the reference's parameter paths and link
 * "output.weight" to "tok_embeddings.weight" (tied head). */
mc_status mc_document_adapt(mc_document* d, int32_t flavour);
mc_status mc_document_save(const mc_document* d, const char* path);      /* src/safetensor.cc:264-290 */

/* options_serializer::load: fills family, head geometry, layer count, rope / norm constants,
 * attn_scale and max_seq_len (1024, as every reference serializer does) from config.json /
 * params.json text; dim / ffn_dim / vocab when the JSON carries them.  Other fields are left as
 * the caller set them. */
mc_status mc_config_from_json(const char* json_text, int32_t flavour, mc_decoder_config* cfg);
/* Widths the reference takes from the tensors themselves: vocab, dim, ffn_dim, group_size, and
 * n_layers / layer_end when still 0.  Call on an adapted document. */
mc_status mc_config_from_document(const mc_document* d, mc_decoder_config* cfg);
/* serializer.load(document): hands every tensor of the layers this decoder owns to
 * mc_decoder_load_linear / _vector / _lora (BF16 <-> F32 converted to the decoder's T). */
mc_status mc_decoder_load_document(mc_decoder* d, const mc_document* doc, int32_t flavour);
mc_status mc_decoder_get_config(const mc_decoder* d, mc_decoder_config* out);

/* ================================================================================================
 * Part 4 -- text in, text out: the callers either side of the token loop.
 *
 *   text::gpt2_codec                include/metalchat/text/gpt.h, src/gpt.cc:20-100
 *   text::byte_pair_encoder<char>   include/metalchat/text/bpe.h:82-345 (tiktoken token map)
 *   text::regexp / regexp_iterator  include/metalchat/text/regexp.h:24-92, src/regexp.cc:34-178 (PCRE2)
 *   reference::llama3_tokenizer_loader   include/metalchat/reference.h:117-165, src/reference.cc:76-127
 *   huggingface llama3_tokenizer_loader  src/llama.cc:81-112 (tokenizer.json)
 *   token scanners, interpreter     include/metalchat/interpreter.h:60-175,296-374, src/interpreter.cc:84-136
 *
 * Host code, no GPU work of its own; the interpreter drives mc_decoder_prefill / mc_decoder_step.
 * The reference's behaviour is kept where it is peculiar, because token ids are the contract:
 *   - the split pattern is compiled by PCRE2 with NO options (src/regexp.cc:38-41): the subject is
 *     bytes, \p{L} / \p{N} see Latin-1 code points, \s is the C-locale set;
 *   - a piece is cut at the PREVIOUS match end with the length of the new match (src/regexp.cc:146-155);
 *   - a piece that is not a token is merged by visiting segments in rank order and joining a segment
 *     with its right neighbour whenever the concatenation is a token; the LAST byte of the piece never
 *     gets a segment of its own and is dropped unless a merge absorbs it (bpe.h:120-168);
 *   - gpt2_codec::decode keeps the low byte of a code point that is not in its table (src/gpt.cc:92-96).
 * One deviation: a pattern that matches the empty string makes the reference loop forever; here it
 * is MC_ERR_RUNTIME "regexp_iterator: empty match".
 * The regular expressions run on the system's libpcre2-8.so.0 (the engine the reference links),
 * opened at first use; without it mc_tokenizer_* fail with MC_ERR_RUNTIME.
 * ============================================================================================== */
typedef struct mc_tokenizer mc_tokenizer;
typedef struct mc_interpreter mc_interpreter;

/* text::token kinds -- include/metalchat/text/tokenizer.h:27-38 */
enum {
    MC_TOKEN_REGULAR = 1 << 0,
    MC_TOKEN_BEGIN_TEXT = 1 << 1,
    MC_TOKEN_END_TEXT = 1 << 2,
    MC_TOKEN_RESERVED = 1 << 3,
    MC_TOKEN_FINETUNE_RIGHT_PAD = 1 << 4,
    MC_TOKEN_BEGIN_HEADER = 1 << 5,
    MC_TOKEN_END_HEADER = 1 << 6,
    MC_TOKEN_END_MESSAGE = 1 << 7,
    MC_TOKEN_END_TURN = 1 << 8,
    MC_TOKEN_IPYTHON = 1 << 9
};

/* Output convention of the byte / id producers below: up to `cap` elements are written, *n always
 * receives the full count (call again with a larger buffer when *n > cap). */

/* gpt2_codec::encode / decode: bytes <-> the printable code points of GPT-2's byte alphabet, UTF-8.
 * decode of malformed UTF-8 or of a code point above U+FFFF: MC_ERR_RUNTIME (std::range_error in the
 * reference's std::wstring_convert). */
mc_status mc_gpt2_encode(const char* bytes, size_t len, char* out, size_t cap, size_t* n);
mc_status mc_gpt2_decode(const char* utf8, size_t len, char* out, size_t cap, size_t* n);

/* The split pattern on its own (text::regexp::begin .. end): piece i occupies
 * [offsets[2i], offsets[2i+1]) of the subject, cut the way the reference's iterator cuts it. */
mc_status mc_regexp_split(const char* pattern, const char* subject, size_t len, size_t* offsets,
                          size_t cap_pairs, size_t* n_pairs);

/* byte_pair_encoder(token_regex): an empty encoder.  token_regex == NULL: the llama3 pattern
 * (reference::llama3_tokenizer_loader::default_regex). */
mc_status mc_tokenizer_create(const char* token_regex, mc_tokenizer** out);
/* reference::llama3_tokenizer_loader::load(path[, regex]): "base64 rank" lines (tiktoken), then the
 * eleven llama3 control tokens at the ids that follow (128000 .. 128010 for the published model). */
mc_status mc_tokenizer_open_tiktoken(const char* path, const char* token_regex, mc_tokenizer** out);
/* huggingface llama3_tokenizer_loader::load(path): tokenizer.json -- the Split pattern of the
 * pre_tokenizer Sequence, model.vocab with GPT-2-coded keys, then the same control tokens. */
mc_status mc_tokenizer_open_hf(const char* tokenizer_json_path, mc_tokenizer** out);
/* text::sentence_piece (include/metalchat/text/sentence_piece.h:17-104): byte-pair merging over the CODE POINTS of the whole
 * text, spaces as U+2581 on the way in and back on the way out.  _create_sentence_piece: an empty one;
 * _open_hf_gemma3: huggingface::gemma3_tokenizer_loader::load (src/gemma.cc:72-94) -- tokenizer.json's model.vocab as
 * spelled, then added_tokens, each bound to a token kind equal to its id.  One difference, where the reference does not
 * return: its ".*" stops at a line feed and its iterator then never advances; here a line is a piece and every line feed a
 * piece of its own. */
mc_status mc_tokenizer_create_sentence_piece(mc_tokenizer** out);
mc_status mc_tokenizer_open_hf_gemma3(const char* tokenizer_json_path, mc_tokenizer** out);
void mc_tokenizer_release(mc_tokenizer* t);
mc_status mc_tokenizer_insert(mc_tokenizer* t, const char* bytes, size_t len, int32_t id, int32_t kind);
mc_status mc_tokenizer_insert_back(mc_tokenizer* t, const char* bytes, size_t len, int32_t kind);
size_t mc_tokenizer_size(const mc_tokenizer* t);
/* encode(string): split, look every piece up, merge the ones that are not tokens. */
mc_status mc_tokenizer_encode(const mc_tokenizer* t, const char* text, size_t len, int32_t* ids,
                              size_t cap, size_t* n);
/* encode(tokenkind): MC_ERR_INVALID_ARGUMENT "byte_pair_encoder: unknown control token '<kind>'" */
mc_status mc_tokenizer_encode_control(const mc_tokenizer* t, int32_t kind, int32_t* id);
/* decode(id): MC_ERR_RUNTIME "byte_pair_encoder: unable to decode id '<id>'" */
mc_status mc_tokenizer_decode(const mc_tokenizer* t, const int32_t* ids, size_t n_ids, char* out,
                              size_t cap, size_t* n);

/* interpreter(transformer, tokenizer): the buffer starts with begin_text, the scanner is
 * limit_token_scanner(50), start_pos 0.  The decoder must own every layer (single stage); its sampler
 * is whatever mc_decoder_set_sampler selected.  Neither handle is owned.  d == NULL gives an interpreter
 * that frames messages (write / pending) but cannot read. */
mc_status mc_interpreter_create(mc_decoder* d, const mc_tokenizer* t, mc_interpreter** out);
void mc_interpreter_release(mc_interpreter* it);
/* set_token_scanner(composite_token_scanner<Op>{limit_token_scanner(limit), match_token_scanner(stop_ids)}):
 * generation continues while op(limit.scan(tok), match.scan(tok)) holds -- op_and != 0:
 * std::logical_and (stop at a stop id OR at the limit, the reference's test set-up), 0: std::logical_or.
 * limit == 0: no limit scanner; n_stop == 0: no match scanner; neither: the empty composite, which
 * stops at once (interpreter.h:148-152). */
mc_status mc_interpreter_set_scanner(mc_interpreter* it, size_t limit, const int32_t* stop_ids,
                                     size_t n_stop, int32_t op_and);
/* declare_variable: "{{ name }}" in later message contents is replaced by value (the reference
 * renders contents with mustache; sections, commands and tool dispatch are not provided). */
mc_status mc_interpreter_declare_variable(mc_interpreter* it, const char* name, const char* value);
/* write(basic_message(role, content)): header, content, end_turn appended to the token buffer. */
mc_status mc_interpreter_write(mc_interpreter* it, const char* role, const char* content);
/* read(): assistant header, flush the buffer through the prompt pass, then one token per step until
 * the scanner says stop; the decoded text goes to out, the ids (optionally) to ids.  sliding_window
 * as for mc_decoder_prefill. */
mc_status mc_interpreter_read(mc_interpreter* it, int32_t sliding_window, char* out, size_t cap, size_t* n,
                              int32_t* ids, size_t ids_cap, size_t* n_ids);
size_t mc_interpreter_start_pos(const mc_interpreter* it);
/* the token buffer that the next read() will flush (for tests) */
mc_status mc_interpreter_pending(const mc_interpreter* it, int32_t* ids, size_t cap, size_t* n);

#ifdef __cplusplus
}
#endif
#endif
