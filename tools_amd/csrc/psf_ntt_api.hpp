// psf_ntt_api.hpp -- what psf_ntt.hip (the translation unit of the NTT kernels) offers the rest of the library.  Device pointers, the caller's
// stream, no allocation per call: the tables of a (device, q, n) are built at first use and kept.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "../../include/psf_mi355x.h"

namespace psf {

// 0: (q, n) has no negacyclic NTT (q not a prime with 4 | q - 1, q >= 2^31, n not a power of two); 1: the generic LDS kernel; 2: one transform per wave
int ntt_route(uint64_t q, size_t n);
// out = a * b mod (X^n + 1, q), `count` products.  io_bits 64: a uint64 (any value), b int64 (any value), out uint64 in [0, q) -- the layout of
// psf_poly_mul_negacyclic; io_bits 16 (route 2, q < 2^14): a uint16 in [0, q), b int16 in (-q, q), out uint16.
psf_status ntt_polymul_dev(int device, uint64_t q, size_t n, size_t count, const void* d_a, const void* d_b, void* d_out, int io_bits, hipStream_t st);
// route 2 only.  hat: count * n words, the register image of the leaf residues (opaque: only ntt_mul_hat_dev / ntt_ring_fa_dev read it)
psf_status ntt_forward_dev(int device, uint64_t q, size_t n, size_t count, const void* d_a, int io_bits, uint32_t* d_hat, hipStream_t st);
// out = a * b with a given by its image; hat_stride in words between the images of consecutive products (0: one image for all)
psf_status ntt_mul_hat_dev(int device, uint64_t q, size_t n, size_t count, const uint32_t* d_hat, size_t hat_stride, const void* d_b, void* d_out, int io_bits, hipStream_t st);
// u_b = sum_{j < K} a_j * sigma_{b,j}: sigma B rows of K*n int64, u B rows of n uint64 (PSFGPVRing::f_a, gpv_ring.rs:243-247)
psf_status ntt_ring_fa_dev(int device, uint64_t q, size_t n, uint32_t K, const uint32_t* d_hat, const int64_t* d_sigma, uint64_t* d_u, size_t B, hipStream_t st);

}  // namespace psf
