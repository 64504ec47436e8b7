/* liblshm_hip — C ABI of the MI355X (gfx950) implementation of the LSHM
 * cascaded-autoencoder + k-harmonic-means training step.
 *
 * The reference (SarodYatawatta/LSHM @ v2) has no native/FFI boundary: its hot
 * path is Python calling PyTorch ATen ops.  Each entry point below therefore
 * names the reference *call site* whose ATen op(s) it replaces (paths relative
 * to the upstream repo root).  INTEGRATION.md shows the ctypes binding a
 * maintainer would add on the reference side.
 *
 * Conventions
 *  - every function returns 0 on success, a negative LSHM_ERR_* code for a bad
 *    argument / unsupported size / short workspace, or a positive hipError_t;
 *    lshm_last_error_string() describes the last failure of the calling thread;
 *  - all tensor pointers are DEVICE pointers to caller-owned, contiguous fp32
 *    storage (row-major / NCHW unless a leading dimension is given); the library
 *    allocates nothing: scratch is passed in, sized by the *_workspace_floats
 *    queries; scalars returned on the device are double precision;
 *  - `stream` is a hipStream_t (pass the caller's current stream); calls only
 *    enqueue work (no synchronisation) and are HIP-graph capturable;
 *  - no global mutable state except the mutex-guarded tile-configuration cache
 *    (lshm_set_tuning / lshm_tuning_import); operand precision is per call (`_bf16`
 *    suffix) or per engine (lshm_step_config.precision); distinct streams may be
 *    driven from distinct threads;
 *  - an engine remembers the HIP device that was current at lshm_engine_create and makes
 *    it current for the duration of every engine call; the per-op entry points run on the
 *    calling thread's current device (the caller's stream must belong to it).
 */
#ifndef LSHM_H
#define LSHM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LSHM_OK 0
#define LSHM_ERR_ARG (-1)
#define LSHM_ERR_WORKSPACE (-2)
#define LSHM_ERR_UNSUPPORTED (-3)
#define LSHM_ERR_COMM (-4)

/* convolution flavours of the autoencoders */
#define LSHM_CONV2D_K4S2P1 0  /* nn.Conv2d(..,4,stride=2,padding=1)          src/lofar_models.py:31-41 */
#define LSHM_TCONV2D_K4S2P1 1 /* nn.ConvTranspose2d(..,4,stride=2,padding=1) src/lofar_models.py:52-57 */
#define LSHM_CONV1D_K4S4P1 2  /* nn.Conv1d(..,4,stride=4,padding=1)          src/lofar_models.py:115-125 */
#define LSHM_TCONV1D_K4S4P0 3 /* nn.ConvTranspose1d(..,4,stride=4,padding=0) src/lofar_models.py:137-142 */

typedef void* lshm_stream_t; /* hipStream_t */

int lshm_version(void);
const char* lshm_last_error_string(void);
/* GEMM-shaped kernels pick their tile configuration per problem shape: from the table of measured shapes
 * (lshm_tuning_import; the one measured on MI355X ships with the package), else a static heuristic -- no
 * timing, no synchronisation, the same kernels in every process (mode 0, the default).  mode 1 (or env
 * LSHM_TUNE=1) is the measuring mode: the table is cleared and the first eager call of each shape times the
 * candidates on the caller's buffers (this synchronises inside the call) and caches the winner.  force >= 0
 * pins configuration `force` (0..21) for every launch (parity tests sweep it); -1 unpins. */
void lshm_set_tuning(int mode, int force);
/* Operand precision of the GEMM-shaped kernels (conv2-5 / tconv0-3 of the three autoencoders, their
 * weight gradients, the dense layers, the dictionary-learning GEMMs) is chosen per call: the plain entry
 * points use fp32 operands on v_mfma_f32_16x16x4_f32 (bitwise an fmaf chain); their `_bf16` forms (below,
 * next to each family) round the operands to bf16 (nearest even) as they are staged in LDS and multiply
 * on v_mfma_f32_16x16x16_bf16 with fp32 accumulation -- BASELINE.json configs[2].  An engine takes its
 * precision from lshm_step_config.precision. */
#define LSHM_PRECISION_F32 0
#define LSHM_PRECISION_BF16_OPERANDS 1
/* engines only: bf16 operands as above AND bf16 STORAGE of the image-sized tensors of the bandwidth-bound part
 * of the step -- the three reconstructions, the row / column residuals fed to the 1-D autoencoders and every
 * image-sized gradient (of the reconstructions, of the residuals, of the 2-D autoencoder's output) -- in the
 * outermost convolution layers and the glue passes that read and write them; fp32 accumulation, fp32 master
 * weights, losses, multipliers and optimiser (BASELINE.json configs[2]) */
#define LSHM_PRECISION_BF16_STORAGE 2
/* The cache as text ("policy M N K Z groups config" per line).  export returns the buffer size needed
 * (terminating 0 included) and fills buf up to cap; import merges entries and returns how many it read.
 * Importing the table measured on the target GPU makes runs start without timing launches and
 * reproduce each other bit for bit. */
size_t lshm_tuning_export(char* buf, size_t cap);
int lshm_tuning_import(const char* text);

/* ---- harmonic features: kron(scales, uv) -> cat(sin, cos)   src/lofar_models.py:60-62,145-147
 * uv (B,2), scales (H) -> out (B,4H) */
int lshm_uv_harmonics(const float* uv, const float* scales, int H, int B, float* out,
                      lshm_stream_t stream);

/* ---- convolution layers, forward + fused bias + optional ELU     src/lofar_models.py:73-78,93-98,158-163,178-183
 * x (B,Cin,Hin,Win) [1D: Hin=1, Win=L]; weight in torch layout (Conv: (Cout,Cin,k..), ConvTranspose:
 * (Cin,Cout,k..)); in_bs / out_bs are batch strides in elements (0 = dense). act: 0 none, 1 ELU.
 * Size limit: the GEMM-shaped layers (and the dense layers below) address each operand tensor with 32-bit
 * byte offsets; a tensor whose last element lies 4 GiB or more past its base pointer is refused with
 * LSHM_ERR_UNSUPPORTED (BASELINE.json's largest configuration stays below 1 GiB per tensor). */
/* workspace floats that let fwd / dgrad / wgrad of one layer use split-K (deep layers) */
size_t lshm_conv_workspace_floats(int kind, int B, int Cin, int Cout, int Hin, int Win);
/* `workspace` may be NULL for fwd / dgrad (no split-K: slower on the deep, few-position layers) */
int lshm_conv_fwd(int kind, const float* x, const float* w, const float* bias, float* y, int B,
                  int Cin, int Cout, int Hin, int Win, long in_bs, long out_bs, int act,
                  float* workspace, size_t workspace_floats, lshm_stream_t stream);
/* Two independent problems of identical shape in one launch: the row- and the column-vectorised 1-D
 * autoencoder (src/kharmonic_lofar.py:142-147) run every layer this way.  A workspace, if given, is
 * split in two: pass twice lshm_conv_workspace_floats(). */
int lshm_conv_fwd_pair(int kind, const float* x0, const float* w0, const float* bias0, float* y0,
                       const float* x1, const float* w1, const float* bias1, float* y1, int B, int Cin,
                       int Cout, int Hin, int Win, long in_bs, long out_bs, int act, float* workspace,
                       size_t workspace_floats, lshm_stream_t stream);
/* data gradient: dx = op^T(dz, w); if y_in_saved != NULL the result is multiplied by ELU'(y_in_saved)
 * (the saved *output* of the previous layer), i.e. it is already the pre-activation gradient. */
int lshm_conv_dgrad(int kind, const float* dz, const float* w, float* dx, const float* y_in_saved,
                    int B, int Cin, int Cout, int Hin, int Win, long in_bs, long out_bs,
                    float* workspace, size_t workspace_floats, lshm_stream_t stream);
/* weight + bias gradient (deterministic split-K); dw/db overwritten unless accumulate != 0;
 * workspace is mandatory here (lshm_conv_workspace_floats) */
int lshm_conv_wgrad(int kind, const float* x, const float* dz, float* dw, float* db, int B, int Cin,
                    int Cout, int Hin, int Win, long in_bs, long out_bs, float* workspace,
                    size_t workspace_floats, int accumulate, lshm_stream_t stream);
int lshm_conv_fwd_bf16(int kind, const float* x, const float* w, const float* bias, float* y, int B,
                       int Cin, int Cout, int Hin, int Win, long in_bs, long out_bs, int act,
                       float* workspace, size_t workspace_floats, lshm_stream_t stream);
int lshm_conv_dgrad_bf16(int kind, const float* dz, const float* w, float* dx, const float* y_in_saved,
                         int B, int Cin, int Cout, int Hin, int Win, long in_bs, long out_bs,
                         float* workspace, size_t workspace_floats, lshm_stream_t stream);
int lshm_conv_wgrad_bf16(int kind, const float* x, const float* dz, float* dw, float* db, int B, int Cin,
                         int Cout, int Hin, int Win, long in_bs, long out_bs, float* workspace,
                         size_t workspace_floats, int accumulate, lshm_stream_t stream);
/* Weight, bias AND data gradient of one layer from ONE pass over dz and the saved input x (the two calls above
 * read both tensors twice).  Available for the bandwidth-bound outer layers: 1-D k4 s4 conv / transposed conv
 * with 4 <-> 8 and 8 <-> 12 channels (src/lofar_models.py:115-117,140-142 backward) and the 2-D k4 s2 p1
 * transposed convs 8 -> 4 and 12 -> 8 (:56-57 backward) and conv 8 -> 12 (:32 backward); LSHM_ERR_UNSUPPORTED otherwise.  elu_grad != 0: dx is multiplied by
 * ELU'(x) (x is then the saved ELU output feeding the layer).  dw / db / dx are overwritten. */
int lshm_conv_bwd_fused(int kind, const float* x, const float* dz, const float* w, float* dw, float* db, float* dx,
                        int elu_grad, int B, int Cin, int Cout, int Hin, int Win, float* workspace,
                        size_t workspace_floats, lshm_stream_t stream);
/* The same under a schedule word (LSHM_SCHED_*, below): LSHM_SCHED_NO_BWD_LDS_8_4 selects the register form of 1-D conv0's
 * kernel (the one bf16 storage runs), LSHM_SCHED_NO_BWD_LDS / _NO_BWD_LDS2D / _NO_BWD_FUSED2D make the call refuse the layers
 * those kernels serve -- what an engine with that word in lshm_step_config.schedule runs. */
int lshm_conv_bwd_fused_ex(int kind, const float* x, const float* dz, const float* w, float* dw, float* db, float* dx,
                           int elu_grad, int B, int Cin, int Cout, int Hin, int Win, float* workspace,
                           size_t workspace_floats, unsigned schedule, lshm_stream_t stream);
/* Three consecutive k4 s4 1-D layers of AutoEncoder1DCNN's middle as ONE launch, the patch's activations resident in
 * LDS from layer to layer (every layer's output is still written to out[k], once, coalesced):
 *   up == 0: stride-4 conv direction, 12 -> 24 -> 48 -> 96 channels from 1024 positions: conv2 -> conv3 -> conv4
 *            forward (pad = 1, act = 1; src/lofar_models.py:119-123) or the data gradients of tconv3 <- tconv2 <- tconv1
 *            (pad = 0, act = 0, dact[k] = the saved input of that layer; :138-140 backward);
 *   up != 0: transposed direction, 96 -> 48 -> 24 -> 12 channels from 16 positions: tconv1 -> tconv2 -> tconv3 forward
 *            (pad = 0, act = 1) or the data gradients of conv4 <- conv3 <- conv2 (pad = 1, act = 0, dact[k]).
 * w[k] / bias[k] (bias may be NULL) in the layers' own torch layouts; dact (NULL or 3 pointers): out[k] *= ELU'(dact[k]). */
int lshm_conv1d_chain3(int up, const float* x, const float* const* w, const float* const* bias, float* const* out,
                       const float* const* dact, int act, int pad, int B, lshm_stream_t stream);
/* The dense middle of AutoEncoder1DCNN(latent_dim=16, rica=True) as ONE launch per direction (src/lofar_models.py:
 * 127-135,165-176 and their backward): 16 batch rows per workgroup stay in LDS through the four layers.
 *   forward: cat1 (B,784) = [conv5 output | elu(fcuv1(uvh))] -> z1 = elu(fc1) (B,16) -> mu = elu(fc2in) (B,16 inside a
 *     (B,ldmu) matrix: the shared latent buffer) -> cat3[:, :16] = elu(fc2out) (cat3 (B,32); its columns 16..31 =
 *     elu(fcuv3(uvh)) must already be there) -> d0 = fc3(cat3) (B,768).  wb = {fc1.weight, fc1.bias, fc2in.weight,
 *     fc2in.bias, fc2out.weight, fc2out.bias, fc3.weight, fc3.bias} (torch layouts).
 *   backward: dd0 (B,768) = gradient of fc3's output -> dcat3 (B,32), dzmu (B,16) (with the latent-term gradient gmu
 *     added before the ELU' multiply), dz1 (B,16), dcat1 (B,784): the pre-activation gradients the weight gradients
 *     and conv5's data gradient read.  w = {fc1.weight, fc2in.weight, fc2out.weight, fc3.weight}. */
int lshm_dense1d_fwd(const float* cat1, const float* const* wb, float* z1, float* mu, long ldmu, float* cat3, float* d0, int B,
                     lshm_stream_t stream);
int lshm_dense1d_bwd(const float* dd0, const float* cat3, const float* mu, long ldmu, const float* gmu, long ldgmu, const float* z1,
                     const float* cat1, const float* const* w, float* dcat3, float* dzmu, float* dz1, float* dcat1, int B,
                     lshm_stream_t stream);
/* The whole mid + deep section of AutoEncoder1DCNN(latent_dim=16, rica=True)'s forward as ONE launch (src/lofar_models.py:
 * 119-135,137-140 and the forward :158-183): conv2 -> conv3 -> conv4 -> conv5 -> fc1 -> fc2in -> fc2out -> fc3 -> tconv0 -> tconv1 ->
 * tconv2 -> tconv3, one workgroup per patch, activations resident in LDS.  x1 (B,12,1024) = conv1's output; w[12] / bias[12] in
 * that order, torch layouts; out[12] = {conv2 (B,24,256), conv3 (B,48,64), conv4 (B,96,16), cat1 (B,784: columns 0..767 written,
 * 768..783 = elu(fcuv1(uvh)) must be there), z1 (B,16), mu (B,16 inside a (B,ldmu) matrix), cat3 (B,32: columns 0..15 written,
 * 16..31 = elu(fcuv3(uvh)) must be there), d0 (B,768), tconv0 (B,96,16), tconv1 (B,48,64), tconv2 (B,24,256), tconv3 (B,12,1024)}.
 * stamps: diagnostics, may be NULL (64 device int64). */
int lshm_chain1d_full_fwd(const float* x1, const float* const* w, const float* const* bias, float* const* out, long ldmu, int B,
                          long long* stamps, lshm_stream_t stream);
/* ---- diagnostics: a per-launch trace of the calling thread WITHOUT a profiler.  Between lshm_trace_begin and lshm_trace_end
 * every kernel this thread launches through the library carries a start and a stop event of the trace's own
 * (hipExtLaunchKernel: the kernel's dispatch and completion timestamps, no marker packets), so the timeline of the
 * SHIPPED schedule can be read where rocprofv3 makes the host the bottleneck.  lshm_trace_end returns the number of
 * launches recorded; after the caller has synchronised the device, lshm_trace_read returns launch `index`: demangled
 * kernel name, start (us after the start of the first recorded launch; a launch on another stream may precede it), duration (us), stream (numbered in order of first use) and
 * grid size in threads.  lshm_trace_free releases the events.  Cost: ~25 launches per iteration that signal a dependency
 * through their stop event record a marker instead. */
int lshm_trace_begin(int capacity);
/* with_start == 0: stop events only -- no start timestamp packet in front of the kernels, so the queues run as they do untraced;
 * lshm_trace_read then returns the COMPLETION time in `start_us` and -1 as the duration. */
int lshm_trace_begin_ex(int capacity, int with_start);
int lshm_trace_end(void);
int lshm_trace_read(int index, char* name, int name_cap, float* start_us, float* dur_us, int* stream_index, unsigned* grid_threads);
int lshm_trace_free(void);
/* The deep section of AutoEncoderCNN2(latent_dim=224, rica=True)'s forward as ONE launch (src/lofar_models.py:36-41,43-51,
 * 52-55 and the forward :66-69,73-98): conv3 -> conv4 -> conv5 -> fc1 -> fc2in -> fc2out -> fc3 -> tconv0 -> tconv1 -> tconv2 ->
 * tconv3, a workgroup per patch (or two), activations resident in LDS, every layer's output also written once to out[]:
 *   x2 (B,24,16,16) = conv2's output; w[11] / bias[11] = {conv3, conv4, conv5, fc1, fc2in, fc2out, fc3, tconv0, tconv1, tconv2,
 *   tconv3} in the layers' own torch layouts; out[11] = {conv3 (B,48,8,8), conv4 (B,96,4,4), cat1 (B,784: columns 0..767
 *   written, 768..783 = elu(fcuv1(uvh)) must be there), z1 (B,224), mu (B,224 inside a (B,ldmu) matrix), cat3 (B,240: columns
 *   0..223 written, 224..239 = elu(fcuv3(uvh)) must be there), d0 (B,768), tconv0 (B,96,4,4), tconv1 (B,48,8,8), tconv2
 *   (B,24,16,16), tconv3 (B,12,32,32)}.
 * The weights are first re-ordered into `packed` (lshm_deep2d_packed_floats() floats, 16-byte aligned) in the order the
 * kernel's matrix instructions consume them (one launch).  variant 0: one patch per 1024-thread workgroup, 1: two patches,
 * 2: one patch per 512-thread workgroup.  Results do not depend on the variant bit for bit.  stamps (diagnostics, may be
 * NULL): 32 device int64 that receive workgroup 0's shader-clock readings at the stage boundaries. */
size_t lshm_deep2d_packed_floats(void);
int lshm_deep2d_fwd(const float* x2, const float* const* w, const float* const* bias, float* const* out, long ldmu, float* packed,
                    int B, int variant, long long* stamps, lshm_stream_t stream);
/* The data-gradient pass back through the same layers (+ conv2's) as ONE launch: the same eleven-stage pipeline on the
 * layers' own weight tensors (the data gradient of a k4 s2 p1 transposed conv is the conv with the same tensor, and vice
 * versa; the dense layers are read transposed).  g_t2 (B,24,16,16) = gradient w.r.t. tconv2's pre-activation output;
 * w[12] = {conv2, conv3, conv4, conv5, fc1, fc2in, fc2out, fc3, tconv0, tconv1, tconv2, tconv3}.weight;
 * saved[10] = the forward tensors whose ELU' multiplies each stage's result: {tconv1 out, tconv0 out, cat3 (B,240), mu
 * (B,224, row pitch ldmu), z1, cat1 (B,784), conv4 out, conv3 out, conv2 out, conv1 out (B,12,32,32)}; gmu (row pitch ldgmu,
 * may be NULL) = gradient of the latent-space terms w.r.t. the code, added in front of fc2out's ELU';
 * out[11] = the pre-activation gradients the weight gradients read as dz: {tconv1 (B,48,8,8), tconv0 (B,96,4,4), fc3's
 * output (B,768), [fc2out | fcuv3] (B,240), fc2in (B,224), fc1 (B,224), [conv5 | fcuv1] (B,784), conv4 (B,96,4,4), conv3
 * (B,48,8,8), conv2 (B,24,16,16), conv1 (B,12,32,32)}.  variant 0: one patch per workgroup, 1: two. */
int lshm_deep2d_bwd(const float* g_t2, const float* const* w, const float* const* saved, long ldmu, const float* gmu, long ldgmu,
                    float* const* out, float* packed, int B, int variant, lshm_stream_t stream);
/* dz = gy * ELU'(y) from the saved output y                      (autograd of F.elu) */
int lshm_elu_bwd(const float* gy, const float* y, float* dz, long n, lshm_stream_t stream);

/* ---- dense layers (F.linear + optional ELU)                    src/lofar_models.py:80-83,89-91,67-68 */
size_t lshm_linear_workspace_floats(int B, int K, int N);
int lshm_linear_fwd(const float* x, long ldx, const float* w, const float* bias, float* y, long ldy,
                    int B, int K, int N, int act, float* workspace, size_t workspace_floats,
                    lshm_stream_t stream);
int lshm_linear_dgrad(const float* dz, long lddz, const float* w, float* dx, long lddx,
                      const float* x_saved, long ldxs, int B, int K, int N, float* workspace,
                      size_t workspace_floats, lshm_stream_t stream);
int lshm_linear_wgrad(const float* x, long ldx, const float* dz, long lddz, float* dw, float* db,
                      int B, int K, int N, float* workspace, size_t workspace_floats,
                      lshm_stream_t stream);

int lshm_linear_fwd_bf16(const float* x, long ldx, const float* w, const float* bias, float* y, long ldy,
                         int B, int K, int N, int act, float* workspace, size_t workspace_floats,
                         lshm_stream_t stream);
int lshm_linear_dgrad_bf16(const float* dz, long lddz, const float* w, float* dx, long lddx,
                           const float* x_saved, long ldxs, int B, int K, int N, float* workspace,
                           size_t workspace_floats, lshm_stream_t stream);
int lshm_linear_wgrad_bf16(const float* x, long ldx, const float* dz, long lddz, float* dw, float* db,
                           int B, int K, int N, float* workspace, size_t workspace_floats,
                           lshm_stream_t stream);

/* ---- K-harmonic means                                           src/lofar_models.py:199-212
 * X (N,D) with leading dimension ldx, M (K,D).  loss_sum[0] = sum_i K/(e_i+eps) (caller divides by
 * N_total*K*D); dX/dM are gradients of gscale * loss_sum * inv_count (inv_count = 1/(N_total*K*D)). */
size_t lshm_khm_workspace_floats(int N, int D, int K);
int lshm_khm_fwd_bwd(const float* X, long ldx, const float* M, int N, int D, int K, float p,
                     float eps, double inv_count, float gscale, double* loss_sum, float* dX,
                     long lddx, float* dM, int accumulate_dx, float* workspace,
                     size_t workspace_floats, lshm_stream_t stream);
/* Zhang's generalised-KHM recursion partial sums (intent of Kmeans.offline_update,
 * src/lofar_models.py:231-261): num (K,D), den (K); M_new = num/den after a sum over ranks. */
int lshm_khm_offline_partials(const float* X, long ldx, const float* M, int N, int D, int K, float p,
                              float eps, float* num, float* den, float* workspace,
                              size_t workspace_floats, lshm_stream_t stream);
/* dist[k] = mean_n ||X_n - M_k||^p                              src/evaluate_clustering.py:111-115 */
int lshm_khm_mean_distances(const float* X, long ldx, const float* M, int N, int D, int K, float p,
                            float* dist, float* workspace, size_t workspace_floats,
                            lshm_stream_t stream);

/* epilogues of that distance vector (K <= 64), either output may be NULL:
 * argmin[0] (device int) = index of the smallest distance, the cluster id of a baseline
 *                          (torch.min(dist.view(Kc,1),0), src/evaluate_clustering.py:116-119);
 * prob[K] = softmax(-dist / mean(dist)), the node labels of src/train_graph_stat.py:206-210 */
int lshm_khm_assign(const float* dist, int K, int* argmin, float* prob, lshm_stream_t stream);

/* ---- Kmeans.cluster_similarity                                  src/lofar_models.py:214-229 */
int lshm_cluster_sim_fwd_bwd(const float* M, int K, int D, float eps, float gscale, double* loss,
                             float* dM, int accumulate, lshm_stream_t stream);
/* ---- augmented_loss(mu, bpb, batch_size)                        src/kharmonic_lofar.py:97-110
 * loss must have room for 1 + ceil(rows/bpb) doubles (loss[0] = result). */
int lshm_aug_loss_fwd_bwd(const float* Z, long ldz, int rows, int D, int bpb, int batch_size,
                          float gscale, double* loss, float* dZ, long lddz, int accumulate,
                          lshm_stream_t stream);
/* ---- RICA penalty scale*sum(log cosh z), dz (+)= scale*tanh z   src/kharmonic_lofar.py:169-171 */
int lshm_logcosh_fwd_bwd(const float* z, long ldz, int rows, int cols, float scale, double* loss,
                         float* dz, long lddz, int accumulate, lshm_stream_t stream);

/* ---- dictionary learning X ~ A S                               src/rica_lofar.py:53-97
 * Everything transposed (patch-major, as the loader delivers it): Xt = x.view(-1, L) (B,L), A (L,M),
 * St = S^T (B,M); all contiguous.  workspace >= lshm_rica_workspace_floats(B, L, M) floats.
 * loss_grad replaces the closure (:72-81): loss[0] = ||X - A S||^2/(B L) + lambda1 ||S||_1/(M B), where
 * ||S||_1 is torch.linalg.norm(S, 1) of a matrix, the largest column sum of |S|; dSt = d loss / d St
 * (NULL: loss only, the closure under no_grad).  update_dictionary replaces :84-93: E = X - A S,
 * A += eta E S^T / B; dA_norm_sq[0] = ||E S^T||_F^2 (the logged ||dA|| is its root over B), may be NULL. */
size_t lshm_rica_workspace_floats(int B, int L, int M);
int lshm_rica_loss_grad(const float* Xt, const float* A, const float* St, int B, int L, int M,
                        float lambda1, double* loss, float* dSt, float* workspace,
                        size_t workspace_floats, lshm_stream_t stream);
int lshm_rica_update_dictionary(const float* Xt, float* A, const float* St, int B, int L, int M,
                                float eta, double* dA_norm_sq, float* workspace,
                                size_t workspace_floats, lshm_stream_t stream);

int lshm_rica_loss_grad_bf16(const float* Xt, const float* A, const float* St, int B, int L, int M,
                             float lambda1, double* loss, float* dSt, float* workspace,
                             size_t workspace_floats, lshm_stream_t stream);
int lshm_rica_update_dictionary_bf16(const float* Xt, float* A, const float* St, int B, int L, int M,
                                     float eta, double* dA_norm_sq, float* workspace,
                                     size_t workspace_floats, lshm_stream_t stream);

/* ---- glue of the closure                                        src/kharmonic_lofar.py:137-158 */
/* out_row = (x-x1)/2, out_col = per-plane transpose of it (planes = B*C planes of P x P) */
int lshm_residual_split(const float* x, const float* x1, float* out_row, float* out_col, int planes,
                        int P, lshm_stream_t stream);
int lshm_plane_transpose(const float* in, float* out, int planes, int P, lshm_stream_t stream);
/* netT.conv0 and netF.conv0 (Conv1d(4, 8, 4, stride=4, padding=1) + ELU, src/lofar_models.py:115) of the row- and the
 * column-vectorised residual (x - x1) / 2 (src/kharmonic_lofar.py:142-147) in ONE launch straight from x and x1
 * ((B,4,128,128) each): neither vectorisation is written.  yT, yF: (B, 8, 4096).  Bitwise the results of
 * lshm_residual_split followed by lshm_conv_fwd_pair; for a forward whose activations are not kept. */
int lshm_resid_conv0(const float* x, const float* x1, const float* wT, const float* bT, float* yT, const float* wF,
                     const float* bF, float* yF, int B, lshm_stream_t stream);
/* The same launch for a forward whose activations ARE kept (the closure forward): also writes the two vectorisations
 * (out_row = the residual as the image, out_col = its per-plane transpose; (B,4,128*128) each) that the backward's
 * weight gradients read.  out_col may be NULL (lshm_conv0_bwd_tile reads the row image alone).  Bitwise
 * lshm_residual_split + lshm_conv_fwd_pair. */
int lshm_resid_conv0_keep(const float* x, const float* x1, const float* wT, const float* bT, float* yT, const float* wF,
                          const float* bF, float* yF, float* out_row, float* out_col, int B, lshm_stream_t stream);
/* Backward of netT.conv0 and netF.conv0 (src/lofar_models.py:115) and the gradient w.r.t. the 2-D reconstruction
 * (src/kharmonic_lofar.py:142-147: both read (x - x1) / 2) in ONE pass over image tiles:
 *   dwT, dbT, dwF, dbF (=|+=) the layers' weight / bias gradients, gx1 = gx1p - (dT + dF^T) / 2
 * resid_row: the residual as the image (B,4,128*128) -- the column-vectorised copy is not read; dzT, dzF: (B,8,4096)
 * gradients w.r.t. the layers' pre-activations; gx1p, gx1: (B,4,128,128).  Replaces lshm_conv_bwd_fused on the pair
 * followed by lshm_combine_dx1 (same values up to fp32 summation order). */
size_t lshm_conv0_bwd_tile_workspace_floats(void);
int lshm_conv0_bwd_tile(const float* resid_row, const float* dzT, const float* dzF, const float* wT, const float* wF,
                        const float* gx1p, float* gx1, float* dwT, float* dbT, float* dwF, float* dbF, int B, float* ws,
                        size_t ws_floats, int accumulate, int storage_bf16, lshm_stream_t stream);
/* (storage_bf16 != 0: resid_row, dzT, dzF, gx1p and gx1 are bf16 tensors -- lshm_step_config.precision 2) */
size_t lshm_recon_workspace_floats(int planes, int P);
/* sums7 = [sum e^2, y1.r1, sum r1^2, y2.r2, sum r2^2, y3.r3, sum r3^2]; gradients scaled by 1/n;
 * x3c / gx3c are in the column-vectorised (per-plane transposed) layout of the third AE.
 * gx1_partial, gx2, gx3c all NULL: only the sums (a closure evaluated under no_grad, :133-134). */
int lshm_recon_losses_fwd_bwd(const float* x, const float* x1, const float* x2, const float* x3c,
                              const float* y1, const float* y2, const float* y3, float rho,
                              int planes, int P, double* sums7, float* gx1_partial, float* gx2,
                              float* gx3c, float* workspace, lshm_stream_t stream);
/* The same terms and gradients with the reconstructions of netT / netF NOT read but formed inside the pass from the inputs
 * aT, aF (planes / C samples x 8 channels x P*P/4 positions) of their last layer, ConvTranspose1d(8, C, 4, stride=4) without
 * activation (src/lofar_models.py:142; weights wT / wF (8, C, 4), biases bT / bF (C)): bitwise what lshm_conv_fwd_pair of that
 * layer followed by lshm_recon_losses_fwd_bwd gives.  The step engine uses the multiplier-updating form of this pass behind
 * its no-grad forward, which then stops one layer early. */
int lshm_recon_losses_from_a(const float* x, const float* x1, const float* aT, const float* aF, const float* wT,
                             const float* bT, const float* wF, const float* bF, const float* y1, const float* y2,
                             const float* y3, float rho, int planes, int P, int C, double* sums7, float* gx1_partial,
                             float* gx2, float* gx3c, float* workspace, lshm_stream_t stream);
/* The pass of the paired schedule, with the backward of the last layer of netT / netF (ConvTranspose1d(8, 4, 4, stride=4),
 * src/lofar_models.py:142) inside it: y_k += rho r_k (src/kharmonic_lofar.py:200-202), then with the new multipliers the seven
 * sums and gx1_partial of lshm_recon_losses_from_a (bitwise), and instead of the two other gradient images
 *   daT, daF (B,8,4096) = gradients w.r.t. the layer's inputs aT, aF times ELU'(input);  dwT, dbT, dwF, dbF = the layers'
 *   weight / bias gradients (overwritten)
 * -- what lshm_conv_bwd_fused(kind 3, elu_grad 1) on gx2 / gx3c gives, to fp32 summation order.  P = 128, C = 4, fp32. */
size_t lshm_recon_bwd5_workspace_floats(int B);
int lshm_recon_bwd5(const float* x, const float* x1, const float* aT, const float* aF, const float* wT, const float* bT,
                    const float* wF, const float* bF, float* y1, float* y2, float* y3, float rho, int B, double* sums7,
                    float* gx1_partial, float* daT, float* daF, float* dwT, float* dbT, float* dwF, float* dbF,
                    float* workspace, size_t workspace_floats, int storage_bf16, lshm_stream_t stream);
/* (storage_bf16 != 0: x1, aT, aF, gx1_partial, daT, daF are bf16 tensors; the reconstructions and the two gradient images are
 *  rounded to bf16 where the separate launches stored them) */
/* The backward of that layer ALONE, from the gradient images gx2 / gx3c (B,4,128,128; gx3c per plane transposed) another pass
 * wrote: the tiles, workgroups and summation order of lshm_recon_bwd5, so that a schedule without the fused pass gets the same
 * bits (what the engine runs for this layer whenever the pass has not done it). */
int lshm_tconv5_pair_bwd(const float* gx2, const float* gx3c, const float* aT, const float* aF, const float* wT, const float* wF,
                         float* daT, float* daF, float* dwT, float* dbT, float* dwF, float* dbF, int B, float* workspace,
                         size_t workspace_floats, int storage_bf16, lshm_stream_t stream);
int lshm_combine_dx1(const float* gx1_partial, const float* gT, const float* gFc, float* gx1,
                     int planes, int P, lshm_stream_t stream);
/* y_k += rho * r_k                                                src/kharmonic_lofar.py:200-202 */
int lshm_multiplier_update(const float* x, const float* x1, const float* x2, const float* x3c,
                           float* y1, float* y2, float* y3, float rho, int planes, int P,
                           lshm_stream_t stream);

/* ---- optimiser algebra on flat arenas                           src/kharmonic_lofar.py:92, src/lbfgsnew.py:84-112 */
int lshm_adam_step_flat(float* p, const float* g, float* m, float* v, long n, float lr, float beta1,
                        float beta2, float eps, const int* step_dev, int step_host, float grad_scale,
                        lshm_stream_t stream);
int lshm_axpy_flat(float* y, const float* x, float alpha, long n, lshm_stream_t stream);
int lshm_scale_flat(float* x, float alpha, long n, lshm_stream_t stream);
/* out[0] = a.b (double); workspace >= 512 floats */
int lshm_dot_flat(const float* a, const float* b, long n, double* out, float* workspace,
                  lshm_stream_t stream);
/* out[0] = sum |a_i| (double); workspace >= 512 floats          src/lbfgsnew.py:531 (flat_grad.abs().sum()) */
int lshm_asum_flat(const float* a, long n, double* out, float* workspace, lshm_stream_t stream);
/* out[i] = a[i].b[i] (double), i < count <= 16, in one launch pair (one host read instead of `count`): the curvature
 * products ys, ss, yy of src/lbfgsnew.py:604-624.  a / b: HOST arrays of device pointers.
 * workspace >= lshm_multi_dot_workspace_doubles(count) doubles. */
size_t lshm_multi_dot_workspace_doubles(int count);
int lshm_multi_dot_flat(const float* const* a, const float* const* b, int count, long n, double* out,
                        double* workspace, size_t workspace_doubles, lshm_stream_t stream);
/* L-BFGS search direction (two-loop recursion, src/lbfgsnew.py:632-651) entirely on the device:
 *   q = -grad; for i = m-1..0: al_i = <s_i,q>/<y_i,s_i>, q -= al_i y_i;  q *= h_diag;
 *   for i = 0..m-1: be_i = <y_i,q>/<y_i,s_i>, q += (al_i - be_i) s_i;   d = q
 * y = old_dirs, s = old_stps (HOST arrays of m <= 16 device pointers, oldest first), d must not alias them.
 * 2m + 2 launches and no host round trip (the host loop needs 3m synchronisations per direction); every inner
 * product is a fixed-order two-stage sum in double: results are reproducible bit for bit. */
size_t lshm_lbfgs_direction_workspace_doubles(int m);
int lshm_lbfgs_direction(const float* const* y, const float* const* s, int m, const float* grad, double h_diag,
                         float* d, long n, double* workspace, size_t workspace_doubles, lshm_stream_t stream);

/* ---- FFT feature step: fftn(dims 2,3, ortho) -> fftshift -> cat(Re,Im) -> clamp
 *      Demo.ipynb:169-175, src/lofar_tools.py:24-30.  x (B,C,128,128) -> out (B,2C,128,128) */
int lshm_fft2_ortho_shift_cat_clamp(const float* x, float* out, int B, int C, float clamp,
                                    lshm_stream_t stream);

/* its backward (the notebooks run the step on detached residuals; this makes the FFT second stage trainable):
 * grad_out, out (B,2C,128,128) = gradient w.r.t. and value of the forward's output (the clamp passes the gradient
 * where |out| < clamp); grad_x (B,C,128,128).  workspace >= lshm_fft2_backward_workspace_floats(B, C). */
size_t lshm_fft2_backward_workspace_floats(int B, int C);
int lshm_fft2_backward(const float* grad_out, const float* out, float* grad_x, int B, int C, float clamp,
                       float* workspace, size_t workspace_floats, lshm_stream_t stream);

/* ---- minibatch patch pipeline (tensor half of get_data_minibatch)      src/lofar_tools.py:113-193
 * vis (nb, ntime, nfreq, 4 pol, 2) int8, scale (nb, nfreq, 4) fp32  ->  y (px*py*nb, 4, patch, patch),
 * px = (max(ntime,patch)-patch)/(patch/2)+1, py likewise; patch-major order (:170-173); clamp (:187);
 * if normalize: (y - mean)/std with torch's unbiased std over the whole minibatch (:190-193).
 * mean_std[2] (device doubles) always receives the pre-normalisation mean and std. */
size_t lshm_patches_workspace_floats(void);
int lshm_patches_from_vis(const int8_t* vis, const float* scale, int nb, int ntime, int nfreq, int patch,
                          float clampv, int normalize, float* y, double* mean_std, float* workspace,
                          lshm_stream_t stream);

/* ---- data-parallel collectives: RCCL over xGMI, one rank per GPU, on the caller's stream.
 * The reference is single-process (no torch.distributed, NCCL or MPI call sites); under data
 * parallelism the quantities its one process consumes whole -- the gradients behind loss.backward()
 * (src/kharmonic_lofar.py:175), the logged terms (:176-181) and the centroid numerator / denominator of
 * Kmeans.offline_update (src/lofar_models.py:240-260) -- are SUM all-reduced over ranks instead.
 * RCCL is bound at run time (dlopen): lshm_comm_available() says whether this process has it.
 * Bootstrap: one rank calls lshm_comm_unique_id and ships the 128 bytes to the others by any means (the
 * host side uses torch.distributed's store); every rank then calls lshm_comm_init with the HIP device it
 * will use current. */
#define LSHM_COMM_ID_BYTES 128
typedef struct lshm_comm lshm_comm;
int lshm_comm_available(void);
int lshm_comm_unique_id(char* id128);
int lshm_comm_init(const char* id128, int rank, int world, lshm_comm** out);
void lshm_comm_destroy(lshm_comm* c);
int lshm_comm_rank(const lshm_comm* c);
int lshm_comm_world(const lshm_comm* c);
/* in place: buf[0..n) (float) and tail[0..ntail) (double) become their sums over ranks; one fused launch
 * (the flat gradient arena and the loss-term vector of one closure; either may be empty) */
int lshm_comm_allreduce_flat(lshm_comm* c, float* buf, size_t n, double* tail, size_t ntail,
                             lshm_stream_t stream);

/* The same with num_channels 4 or 8 (8: real / imaginary parts of all four polarisations, :101-111) and
 * with the un-normalised moments exposed: moments[3] (device doubles, may be NULL) = [sum, sum of squares,
 * count] of this call's clamped values.  Upstream normalises by the mean / std of the WHOLE minibatch
 * (:190-193); ranks of a data-parallel job call this with normalize = 0, SUM all-reduce the three doubles
 * and finish with lshm_patches_normalize, which reproduces the single-process global-batch statistics. */
int lshm_patches_from_vis_ex(const int8_t* vis, const float* scale, int nb, int ntime, int nfreq, int patch,
                             int num_channels, float clampv, int normalize, float* y, double* mean_std,
                             double* moments, float* workspace, lshm_stream_t stream);
/* y[0..n) = (y - mean) / std, mean and unbiased std from moments = [sum, sum of squares, count] */
int lshm_patches_normalize(float* y, long n, const double* moments, lshm_stream_t stream);

/* ---- fused training-step engine (one ADMM iteration)            src/kharmonic_lofar.py:131-202 */
typedef struct lshm_engine lshm_engine;
typedef struct lshm_step_config {
  int B, C, P;          /* batch, channels (4), patch size (128) */
  int L, Lt, K;         /* latent dims of AE1 / 1D AEs, number of centroids */
  float p;              /* K-harmonic order */
  float alpha, beta, gamma, rho, rica_lambda;
  int rica;             /* 0/1 */
  int bpb, batch_size;  /* patches per baseline, baselines used by augmented_loss */
  int H;                /* number of harmonic scales (4) */
  float scales[8];
  int world;            /* data-parallel world size (gradients are averaged over it) */
  int precision;        /* LSHM_PRECISION_*: operand precision of this engine's GEMM-shaped layers */
  unsigned schedule;    /* LSHM_SCHED_*: 0 = the shipped schedule; each bit switches ONE of its choices off (per engine: two
                           engines of a process may differ).  Every alternative is the launch sequence the choice replaced;
                           results agree to rounding (bit for bit where noted). */
  unsigned tune;        /* experimental placement word of the round's A/B measurements; 0 = the shipped schedule */
} lshm_step_config;
/* conv3 .. tconv3 of the 2-D autoencoder's forward as eleven launches instead of one (lshm_deep2d_fwd) */
#define LSHM_SCHED_NO_DEEP2D (1u << 0)
/* ... and the data gradients of tconv2 .. conv2 of the 2-D autoencoder as eleven launches instead of one (lshm_deep2d_bwd) */
#define LSHM_SCHED_NO_DEEP2D_BWD (1u << 1)
/* the implicit-GEMM weight gradients of the 2-D autoencoder's deep layers as six launches instead of one batched launch */
#define LSHM_SCHED_NO_WGRAD_BATCH (1u << 2)
/* EXPERIMENT, off unless set: conv2 .. tconv3 of the 1-D autoencoders' forward as ONE launch per pair (chain1d_full.hip) instead of
 * five (two three-layer chains, conv5, the dense middle, tconv0).  Measured slower in the step (profiles/r04/README.md): a
 * workgroup per (patch, network) re-reads conv5's and tconv0's weights from L2 for every patch and runs them on 6-12 of its
 * 16 wavefronts, 55 us per workgroup wave where the five launches overlap with the other forward. */
#define LSHM_SCHED_TRY_FULL1D (1u << 3)
/* The remaining choices, one bit each (what runs instead is the launch sequence the choice replaced): */
#define LSHM_SCHED_NO_CHAIN1D (1u << 4)       /* conv2-4 / tconv1-3 of the 1-D pair's FORWARD as three launches each (lshm_conv1d_chain3) */
#define LSHM_SCHED_NO_CHAIN1D_BWD (1u << 5)   /* ... their data gradients as three launches each */
#define LSHM_SCHED_NO_DENSE1D (1u << 6)       /* the dense middle of the 1-D pair's forward as five launches (lshm_dense1d_fwd) */
#define LSHM_SCHED_NO_DENSE1D_BWD (1u << 7)   /* ... of its backward (lshm_dense1d_bwd) */
#define LSHM_SCHED_NO_RESID_CONV0 (1u << 8)   /* no-grad forward: residual split + conv0 of netT / netF as two launches (lshm_resid_conv0) */
#define LSHM_SCHED_NO_RECON_FROM_A (1u << 9)  /* no-grad forward runs the last 1-D layer; the reconstruction pass reads its output */
#define LSHM_SCHED_NO_ONE_PASS_BWD (1u << 10) /* every one-pass (data + weight + bias gradient) kernel off: separate launches (lshm_conv_bwd_fused) */
#define LSHM_SCHED_NO_BWD_LDS (1u << 11)      /* the LDS-staged one-pass kernels of the 1-D 12/8 layers and of 1-D conv0 off */
#define LSHM_SCHED_NO_BWD_LDS_8_4 (1u << 12)  /* 1-D conv0's one-pass backward on the register form (what bf16 storage uses) */
#define LSHM_SCHED_NO_BWD_LDS2D (1u << 13)    /* the one-pass backward of the 2-D 12/8 layers (tconv4, conv1) off */
#define LSHM_SCHED_NO_BWD_FUSED2D (1u << 14)  /* the one-pass backward of the outermost 2-D decoder layer off */
#define LSHM_SCHED_NO_WGRAD_MID (1u << 15)    /* weight gradients of the mid 1-D layers as implicit GEMMs */
#define LSHM_SCHED_NO_STOP_EVENTS (1u << 16)  /* "dz ready" as recorded events (marker packets) instead of kernel completion signals */
#define LSHM_SCHED_WGRAD_INLINE (1u << 17)    /* weight gradients on the data-gradient stream (no second stream) */
#define LSHM_SCHED_FORK (1u << 18)            /* netT and netF on two streams instead of paired launches */
#define LSHM_SCHED_PHASE_EVENTS (1u << 19)    /* diagnostic: record the phase-boundary events lshm_engine_phase_times reads */
#define LSHM_SCHED_NO_KHM_MFMA (1u << 20)     /* K-harmonic pass for 16 < K <= 64 on the row-split kernel instead of the matrix cores */
#define LSHM_SCHED_NO_EARLY_LATENT (1u << 21) /* latent-space terms at the head of the backward instead of beside the paired forwards */
#define LSHM_SCHED_NO_CONV0_BWD_TILE (1u << 23) /* backward of 1-D conv0 (netT, netF) and the combination into the 2-D autoencoder's output gradient as two launches (lshm_conv0_bwd_tile) */
#define LSHM_SCHED_NO_SHARED_PACK (1u << 25)    /* paired forwards: each chain makes its own fragment-ordered copy of the deep weights */
#define LSHM_SCHED_NO_RECON_BWD5 (1u << 24)     /* the reconstruction pass does not include the backward of netT / netF's last layer (lshm_recon_bwd5) */
#define LSHM_SCHED_NO_RESID_CONV0_KEEP (1u << 22) /* closure forward: residual split + conv0 of netT / netF as two launches (lshm_resid_conv0_keep) */

int lshm_engine_create(const lshm_step_config* cfg, lshm_engine** out);
void lshm_engine_destroy(lshm_engine* e);
/* number of floats in the flat parameter arena and the offset of a named tensor
 * ("net.conv0.weight", "netT.fc1.bias", "mod.M", ...); names follow the reference's state_dict keys */
long lshm_engine_param_count(const lshm_engine* e);
int lshm_engine_param_lookup(const lshm_engine* e, const char* name, long* offset, long* numel);
int lshm_engine_param_name(const lshm_engine* e, int index, char* buf, int buflen, long* offset,
                           long* numel, int* ndim, long* shape /* >= 4 */);
size_t lshm_engine_workspace_floats(const lshm_engine* e);
/* Attach a communicator (NULL detaches): the closures then return GLOBAL gradients and terms, i.e.
 * lshm_engine_forward_backward[_ex] / lshm_engine_backward_saved sum-all-reduce [grads | terms] and
 * lshm_engine_forward_loss the terms, inside the call: the gradients of the two 1-D autoencoders go as soon
 * as their backward is complete, on a stream of their own beside the 2-D autoencoder's backward; the rest
 * follows the last weight gradient.  The communicator's world size must equal lshm_step_config.world. */
int lshm_engine_set_comm(lshm_engine* e, lshm_comm* comm);
/* The early bucket exists only where the engine has its side streams; the SEQUENCE of collectives must be the
 * same on every rank, so the ranks agree on it once: each asks lshm_engine_comm_early_bucket (1: this engine
 * would send the netT / netF gradients early), the answers are MIN-reduced over the job, and the result goes to
 * lshm_engine_set_early_bucket on every rank (0: one closing group only).  A captured call (HIP graph) never uses
 * the early bucket: it would be a fork nested in the forked weight-gradient stream, which hipStreamEndCapture of
 * ROCm 7.2 does not survive. */
int lshm_engine_comm_early_bucket(const lshm_engine* e);
int lshm_engine_set_early_bucket(lshm_engine* e, int on);
/* HIP device the engine was created on (-1: no device): every engine call makes it current, refuses a stream of
 * another device and refuses arena / workspace / input pointers that are not device memory of that device. */
int lshm_engine_device(const lshm_engine* e);
/* Replace the per-call bits of the engine's schedule word (lshm_step_config.schedule; for A/B measurements and the tests
 * that hold a fused kernel to the launches it replaced on ONE set of parameters).  The bits that shape the engine at
 * creation -- LSHM_SCHED_NO_DEEP2D, _NO_DEEP2D_BWD, _TRY_FULL1D (workspace), _WGRAD_INLINE, _FORK, _PHASE_EVENTS (streams,
 * events) -- keep their creation-time value.  Returns the word in effect. */
unsigned lshm_engine_set_schedule(lshm_engine* e, unsigned schedule);
/* Diagnostic (an engine created with LSHM_PHASE_EVENTS=1 in the environment; LSHM_ERR_UNSUPPORTED otherwise): device
 * timestamps at the phase boundaries of the last iteration (lshm_engine_backward_saved, the optimiser, then
 * lshm_engine_multiplier_update_next_ex), in milliseconds after the closure's first launch; synchronises the device.
 * ms[0..9] = closure entry (0), 1-D backward done, main-stream backward done, weight-gradient stream done, closure end,
 * update entry (after the optimiser step), closure forward done, no-grad forward done, reconstruction pass done, update
 * end; -1 for a phase the schedule did not run.  No profiler involved: each mark is one hipEventRecord. */
int lshm_engine_phase_times(const lshm_engine* e, float* ms, int n);
/* what the LAST engine call on `e` actually did (diagnostics, tests) */
#define LSHM_ENGINE_USED_EARLY_BUCKET 1u       /* the netT / netF gradients went as an early all-reduce bucket */
#define LSHM_ENGINE_USED_CONCURRENT_FORWARD 2u /* two forwards ran side by side (LSHM_NEXT_CONCURRENT_FORWARD) */
unsigned lshm_engine_last_flags(const lshm_engine* e);
/* closure forward + backward: fills grads (same layout as params) and terms[16] (double, device):
 * [0..7] = loss0, loss1, loss2, loss3, kdist, aug, sim, rica (already weighted, as logged upstream),
 * [8] = total, [9] = how many of [0..7] are NaN or infinite (0 = healthy; sums over ranks like the rest).  With world > 1 the loss terms / gradients are this rank's share (sum over ranks = global). */
int lshm_engine_forward_backward(lshm_engine* e, const float* params, float* grads, const float* x,
                                 const float* uv, const float* y1, const float* y2, const float* y3,
                                 double* terms, float* workspace, size_t workspace_floats,
                                 lshm_stream_t stream);
/* flags of lshm_engine_forward_backward_ex */
#define LSHM_STEP_RECON_READY 1u /* the last engine call on this workspace was lshm_engine_multiplier_update_next
                                  * with these params / x / uv and the multipliers have not changed since: the
                                  * reconstruction terms and their three gradient images it left behind are the ones
                                  * this closure would compute (bit for bit), so that pass is skipped; the forward
                                  * itself is recomputed, as upstream does (src/kharmonic_lofar.py:135-150) */
int lshm_engine_forward_backward_ex(lshm_engine* e, const float* params, float* grads, const float* x,
                                    const float* uv, const float* y1, const float* y2, const float* y3,
                                    double* terms, float* workspace, size_t workspace_floats, unsigned flags,
                                    lshm_stream_t stream);
/* The same closure on the activations a previous forward left in `workspace`.  Inside the ADMM loop
 * (src/kharmonic_lofar.py:131-202) the no-grad forward of iteration k (lshm_engine_multiplier_update,
 * after the optimizer step) and the closure forward of iteration k+1 evaluate the same networks on the
 * same minibatch with the same parameters - only the multipliers y differ, and they enter after the
 * forward - so iteration k+1 may start from the saved activations: losses, gradients and terms are
 * bit-for-bit those of lshm_engine_forward_backward.  The caller guarantees that the last engine call on
 * this workspace was a forward with these params / x / uv. */
int lshm_engine_backward_saved(lshm_engine* e, const float* params, float* grads, const float* x,
                               const float* y1, const float* y2, const float* y3, double* terms,
                               float* workspace, size_t workspace_floats, lshm_stream_t stream);
/* lshm_engine_multiplier_update that also leaves the reconstruction terms of the *next* closure in the
 * workspace (they read the same seven image-sized arrays as the multiplier update: one pass instead of two).
 * A following lshm_engine_backward_saved, or lshm_engine_forward_backward_ex with LSHM_STEP_RECON_READY, picks
 * them up; any other forward discards them. */
int lshm_engine_multiplier_update_next(lshm_engine* e, const float* params, const float* x, const float* uv,
                                       float* y1, float* y2, float* y3, float* workspace,
                                       size_t workspace_floats, lshm_stream_t stream);
/* flags of lshm_engine_multiplier_update_next_ex */
#define LSHM_NEXT_CONCURRENT_FORWARD 1u /* ALSO run the closure forward of the next iteration, concurrently: the no-grad
                                         * forward that closes iteration k (src/kharmonic_lofar.py:187-196) and the closure
                                         * forward that opens iteration k+1 (:135-150) depend on the updated parameters
                                         * alone, so they are issued as two chains on two HIP streams with separate
                                         * activation buffers (both are computed, as upstream computes both; neither
                                         * waits for the other).  The next closure is then lshm_engine_backward_saved. */
int lshm_engine_multiplier_update_next_ex(lshm_engine* e, const float* params, const float* x, const float* uv,
                                          float* y1, float* y2, float* y3, float* workspace,
                                          size_t workspace_floats, unsigned flags, lshm_stream_t stream);
/* closure forward only (line-search evaluations of LBFGS): terms as above */
int lshm_engine_forward_loss(lshm_engine* e, const float* params, const float* x, const float* uv,
                             const float* y1, const float* y2, const float* y3, double* terms,
                             float* workspace, size_t workspace_floats, lshm_stream_t stream);
/* no-grad forward of the three AEs and y_k += rho r_k              src/kharmonic_lofar.py:187-202 */
int lshm_engine_multiplier_update(lshm_engine* e, const float* params, const float* x,
                                  const float* uv, float* y1, float* y2, float* y3,
                                  float* workspace, size_t workspace_floats, lshm_stream_t stream);
/* forward of the three AEs only: latents Mu (B, L+2Lt) and reconstructions (optional outputs) */
int lshm_engine_encode(lshm_engine* e, const float* params, const float* x, const float* uv,
                       float* Mu, float* x1, float* x2, float* x3, float* workspace,
                       size_t workspace_floats, lshm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* LSHM_H */
