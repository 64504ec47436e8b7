// Launcher interface of chain1d_full.hip: the whole mid + deep section of AutoEncoder1DCNN (conv2 .. tconv3, twelve layers) as
// one launch; [2]: the two problems of a paired launch (netT, netF).
#pragma once
#include "kernels.h"

namespace lshm {

struct Chain1dFullArgs {
  const float* in[2]; long in_bs;                 // conv1's output (B, 12, 1024)
  Chain1dStage dn[3];                             // conv2, conv3, conv4: weights, biases, outputs (act = 1)
  const float *w5[2], *b5[2];                     // conv5 (192, 96, 4)
  float* cat1[2];                                 // (B, 784): columns 0..767 = conv5's output (written), 768..783 = elu(fcuv1(uvh)) (read)
  const float *fc1w[2], *fc1b[2], *fc2inw[2], *fc2inb[2], *fc2outw[2], *fc2outb[2], *fc3w[2], *fc3b[2];
  float* z1[2];                                   // (B, 16)
  float* mu[2]; long mu_ld;                       // (B, 16) inside the shared latent matrix
  float* cat3[2];                                 // (B, 32): columns 0..15 written, 16..31 = elu(fcuv3(uvh)) read
  float* d0[2];                                   // (B, 768) fc3's output
  const float *wt0[2], *bt0[2];                   // tconv0 (192, 96, 4)
  float* t0[2];                                   // tconv0's output (B, 96, 16)
  Chain1dStage up[3];                             // tconv1, tconv2, tconv3
  long long* stamps;                              // diagnostics (or null): shader-clock readings of workgroup (0, 0) at the stage boundaries
};
bool chain1d_full_supported(int L, int hd, int rica, const int* ch, int L1);
int chain1d_full_fwd(const Chain1dFullArgs& a, int B, int nproblems, hipStream_t s);

}  // namespace lshm
