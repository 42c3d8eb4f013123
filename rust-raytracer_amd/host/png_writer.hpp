// Minimal PNG (8-bit RGB, stored deflate blocks) and binary PPM writers for the
// frame the reference saves with the image crate (src/main.rs:181-188).
#pragma once

#include <cstdint>
#include <string>

namespace host {

bool write_png(const std::string &path, const uint8_t *rgb, uint32_t width, uint32_t height);
bool write_ppm(const std::string &path, const uint8_t *rgb, uint32_t width, uint32_t height);

} // namespace host
