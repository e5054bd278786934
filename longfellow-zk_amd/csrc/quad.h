// quad.h -- an uploaded circuit layer (lfgpu_quad) as quad.hip, sumcheck.hip and zk256.hip see it.
#pragma once
#include <memory>
#include <vector>

#include "ctx.h"

#define BG_THREADS 1024  // Quad::bind_g kernels: long runs fold inside the block (runfold.h); d_runoff is per block of this size

struct __attribute__((aligned(16))) corner4 {
  u32 g, h0, h1, vi;
};

// the device arrays of an uploaded layer: immutable after lf_quad_upload_corners, so the quads of several contexts on the
// same device may read them (lf_quad_share: K provers in K host threads, one copy of the circuit)
struct QuadArrays {
  void* p[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  ~QuadArrays();
};

struct lfgpu_quad {
  lfgpu_ctx* c;
  std::shared_ptr<QuadArrays> arrays;  // owner of the six pointers below once the upload has succeeded
  int field;
  size_t n, nk, nv;
  size_t hmax;        // largest hand index (h0 or h1) of any corner: every consumer needs nw > hmax
  corner4* d_morton;  // canonical order
  corner4* d_bygate;  // sorted by g (stable)
  u32* d_goff;        // nv + 1 offsets into d_bygate
  elt_t* d_kvec;      // nk constants (field 1, Fp256Base: nk 32-byte elements behind the same pointer)
  // the run structure of the canonical order (which terms share a hand pair) depends on the circuit only: block offsets of
  // the run heads and the HQUAD size of Quad::bind_g are computed once at upload
  u32* d_runoff;      // per block of BG_THREADS terms: number of run heads before it
  u32* d_nh;          // device copy of nh0
  size_t nh0;
  // the same holds for every HQuad::bind_h of the layer's sumcheck: filled by the first proof for the round-hands that
  // run on the multi-kernel path (per-block output offsets + the size after the bind), reused by every later proof
  struct BindShape {
    u32* d_off;
    size_t n_in, n_out;
  };
  std::vector<BindShape> bind_shape;  // indexed by round-hand
  ScGridOffCache grid_off;            // ... and for the round-hands that run on the shrinking grid
};

// lfgpu_quad_upload for corners that are already packed and range-checked (g < nv, vi < nk; hmax = largest hand index): the
// canonical order goes up once, the by-gate order and its offsets are built on the device (quad.hip)
int lf_quad_upload_corners(lfgpu_ctx* c, int field, size_t n, const corner4* corners, size_t hmax, size_t nk, const void* h_kvec, size_t nv, lfgpu_quad** out);
// a second handle on the same device arrays for another context of the same device (its own stream, scratch and per-proof
// caches: the recorded bind shapes / grid offsets are filled by each handle's first proof)
int lf_quad_share(lfgpu_ctx* c, const lfgpu_quad* src, lfgpu_quad** out);
