// Entry points whose kernels are not written yet: they fail loudly (no CPU stand-in).
#include "chip_internal.h"

namespace chip {
hipError_t launch_deflate_l1(const BatchArgs &, hipStream_t) { return hipErrorNotSupported; }
}  // namespace chip

extern "C" {
chip_encoder *chip_encoder_new(const chip_encoder_opts *) { return nullptr; }
chip_encode_result chip_encode(chip_encoder *, const uint8_t *, size_t in_len, uint8_t *, size_t out_len, int)
{
    chip_encode_result r = {in_len, out_len, CHIP_ENC_ERROR};
    return r;
}
chip_encoder *chip_encoder_reset(chip_encoder *e) { return e; }
void chip_encoder_free(chip_encoder *) {}
int chip_encode_batch(int, int, size_t, const void *, const uint64_t *, const uint32_t *, void *, const uint64_t *,
                      const uint32_t *, uint32_t *, int32_t *, void *)
{
    return CHIP_E_INVALID;
}
size_t chip_encode_bound(int, size_t in_len) { return in_len + (in_len >> 3) + 64; }
}
