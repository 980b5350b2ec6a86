/* Measurement probes for MI355X (gfx950) - NOT part of the product library.
 * Built into the separate libmi355probe.so (csrc/probe.hip, `make probe`); used by scripts/mfma_ceiling.py only. */
#ifndef MI355PROBE_H_
#define MI355PROBE_H_
#ifdef __cplusplus
extern "C" {
#endif
/* blocks x 256 threads run `iters` trips of 48 v_mfma_f32_16x16x32_bf16 per wave on random operands (mode 0: operands in
 * registers; mode 1: plus 18 ds_read_b128 fragment reads per trip, the halo kernel's LDS diet) or 24 v_mfma_f32_32x32x16_bf16
 * (mode 2).  rnd: 1 MiB of finite bf16 bit patterns; s: hipStream_t. */
int mi355_probe_mfma(int mode, const void* rnd, int blocks, int iters, float* sink, void* s);
/* One 1-KiB LDS-DMA piece through a buffer descriptor (dma.hpp: dma16_buf): lane i fetches 16 B at src + soff + 16 i, lanes in
 * `pad_mask` use the always-out-of-range offset, valid == 0 sets num_records = 0.  out[256]: the KiB found in LDS afterwards
 * (pre-filled with 0x7f bytes) — tests/test_gpu_conv.py pins that the range check writes zeros. */
int mi355_probe_bufdma(const void* src, unsigned soff, unsigned long long pad_mask, int valid, unsigned* out, void* s);
#ifdef __cplusplus
}
#endif
#endif
