// frame_bits.h -- device helpers shared by the frame bit writers (frame_pack.cpp and the PACK
// variant of the wave kernel): MSB-first bit placement in a zeroed LDS word buffer, CRC-8 / CRC-16.
#ifndef FLACENC_HIP_FRAME_BITS_H_
#define FLACENC_HIP_FRAME_BITS_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace flacenc_hip {
namespace {

// bit position b of the frame = bit (31 - b % 32) of word b / 32 (words are stored big-endian)
__device__ __forceinline__ void put_bits(uint32_t* w, uint32_t bitpos, uint32_t value, uint32_t nbits) {
  if (nbits == 0) return;
  const uint32_t word = bitpos >> 5, end = (bitpos & 31u) + nbits;  // 1..63: bits used from `word` on
  if (end <= 32u) {
    atomicOr(&w[word], value << (32u - end));
  } else {
    atomicOr(&w[word], value >> (end - 32u));
    atomicOr(&w[word + 1], value << (64u - end));
  }
}

__device__ __forceinline__ uint32_t zigzag32(int32_t v) {  // rice::encode_signbit, rice.rs:169-171
  return ((uint32_t)v << 1) ^ (uint32_t)(v >> 31);
}

// a * b mod (x^16 + x^15 + x^2 + 1) over GF(2)
__device__ __forceinline__ uint32_t gf_mulmod16(uint32_t a, uint32_t b) {
  uint32_t r = 0;
  for (int i = 15; i >= 0; --i) {
    r <<= 1;
    if (r & 0x10000u) r ^= 0x18005u;
    if ((b >> i) & 1u) r ^= a;
  }
  return r & 0xFFFFu;
}

__device__ __forceinline__ uint32_t crc16_byte(uint32_t crc, uint32_t byte) {
  crc ^= byte << 8;
#pragma unroll
  for (int b = 0; b < 8; ++b) crc = (crc & 0x8000u) ? ((crc << 1) ^ 0x8005u) & 0xFFFFu : (crc << 1) & 0xFFFFu;
  return crc;
}

__device__ __forceinline__ uint32_t crc8_byte(uint32_t crc, uint32_t byte) {
  crc ^= byte;
#pragma unroll
  for (int b = 0; b < 8; ++b) crc = (crc & 0x80u) ? ((crc << 1) ^ 0x07u) & 0xFFu : (crc << 1) & 0xFFu;
  return crc;
}

}  // namespace
}  // namespace flacenc_hip
#endif
