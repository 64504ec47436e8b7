// Launcher interface of deep2d.hip (its own header: the kernels are still moving, and kernels.h rebuilds the library).
#pragma once
#include "common.h"

namespace lshm {

// ---- the deep section of AutoEncoderCNN2 as one launch per direction (deep2d.hip): conv3 -> conv4 -> conv5 -> fc1 -> fc2in ->
// fc2out -> fc3 -> tconv0 -> tconv1 -> tconv2 -> tconv3 and the data-gradient pass back through the same layers (+ conv2's),
// G patches per workgroup, activations resident in LDS, weights streamed from a fragment-ordered copy that deep2d_pack
// makes (once per pass: the parameters may have changed)
struct Deep2dWeights {  // the layers' own tensors (torch layouts); c2 is read by the backward packing only
  const float *c2, *c3, *c4, *c5, *fc1, *fc2in, *fc2out, *fc3, *t0, *t1, *t2, *t3;
};
struct Deep2dIO {
  const float* x2;                                                        // conv2 output (B, 24, 16, 16)
  const float *b3, *b4, *b5, *bfc1, *bfc2in, *bfc2out, *bfc3, *bt0, *bt1, *bt2, *bt3;  // biases
  float *a3, *a4;                                                         // conv3 / conv4 outputs
  float* cat1;                                                            // (B, 784): [0, 768) written, [768, 784) read
  float* z1;                                                              // (B, 224)
  float* mu; long mu_ld;                                                  // (B, 224), row pitch mu_ld
  float* cat3;                                                            // (B, 240): [0, 224) written, [224, 240) read
  float* d0;                                                              // (B, 768)
  float *t0, *t1, *t2, *t3;                                               // tconv0..3 outputs
  long long* stamps = nullptr;                                            // diagnostics: 32 shader-clock readings of workgroup 0 (or null)
};
// Data gradients.  g_*: gradient w.r.t. the PRE-activation output of that layer (what its weight gradient reads as dz);
// s_*: the saved forward tensors whose ELU' multiplies them.
struct Deep2dBwdIO {
  const float* g_t2;                         // in: gradient w.r.t. tconv2's pre-activation output (B, 24, 16, 16)
  const float *s_t1, *s_t0;                  // tconv1 / tconv0 outputs (B, 48, 8, 8), (B, 96, 4, 4)
  const float* s_cat3;                       // (B, 240) [elu(fc2out) | elu(fcuv3)]
  const float* s_mu; long s_mu_ld;           // the latent code (B, 224), row pitch s_mu_ld
  const float* gmu; long gmu_ld;             // gradient of the latent-space terms w.r.t. the code (may be null)
  const float* s_z1;                         // (B, 224)
  const float* s_cat1;                       // (B, 784) [conv5 output | elu(fcuv1)]
  const float *s_c4, *s_c3, *s_c2, *s_c1;    // conv4..conv1 outputs (B, 96, 4, 4), (B, 48, 8, 8), (B, 24, 16, 16), (B, 12, 32, 32)
  float *g_t1, *g_t0;                        // out: (B, 48, 8, 8), (B, 96, 4, 4)
  float* g_d0;                               // (B, 768): gradient w.r.t. fc3's output
  float* g_cat3;                             // (B, 240): pre-activation gradients of fc2out (224) and fcuv3 (16)
  float* g_mu; long g_mu_ld;                 // (B, 224): pre-activation gradient of fc2in
  float* g_z1;                               // (B, 224): pre-activation gradient of fc1
  float* g_cat1;                             // (B, 784): pre-activation gradients of conv5 (768) and fcuv1 (16)
  float *g_c4, *g_c3, *g_c2, *g_c1;          // pre-activation gradients of conv4..conv1
  long long* stamps = nullptr;
};
bool deep2d_supported(int L, int hd, int rica, const int* enc_ch, int H2);
size_t deep2d_packed_floats();  // of either direction's copy
int deep2d_pack(const Deep2dWeights& w, float* packed, int backward, int bf16_weights, hipStream_t st);
// variant 0: one patch per 1024-thread workgroup, 1: two patches, 2 (forward only): one patch per 512-thread workgroup;
// + 4: `packed` holds bf16 weights (deep2d_pack(.., bf16_weights = 1)): half the L2 stream, fp32 products and accumulation
int deep2d_fwd(const Deep2dIO& io, const float* packed, int B, int variant, hipStream_t st);
int deep2d_bwd(const Deep2dBwdIO& io, const float* packed, int B, int variant, hipStream_t st);

}  // namespace lshm
