// Launcher interface of deep2d.hip (its own header: the kernels are still moving, and kernels.h rebuilds the library).
#pragma once
#include "common.h"

namespace lshm {

// ---- the deep section of AutoEncoderCNN2's forward as one launch (deep2d.hip): conv3 -> conv4 -> conv5 -> fc1 -> fc2in ->
// fc2out -> fc3 -> tconv0 -> tconv1 -> tconv2 -> tconv3, G patches per workgroup, activations resident in LDS, weights streamed
// from a fragment-ordered copy that deep2d_pack makes (once per forward: the parameters may have changed)
struct Deep2dWeights {  // the layers' own tensors (torch layouts)
  const float *c3, *c4, *c5, *fc1, *fc2in, *fc2out, *fc3, *t0, *t1, *t2, *t3;
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
bool deep2d_supported(int L, int hd, int rica, const int* enc_ch, int H2);
size_t deep2d_packed_floats();
int deep2d_pack(const Deep2dWeights& w, float* packed, hipStream_t st);
// variant 0: one patch per 1024-thread workgroup, 1: two patches, 2: one patch per 512-thread workgroup
int deep2d_fwd(const Deep2dIO& io, const float* packed, int B, int variant, hipStream_t st);

}  // namespace lshm
