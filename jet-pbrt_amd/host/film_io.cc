// jet-pbrt_amd/host/film_io.cc -- FFilm::SaveAsImage (film.h:84, film.cc:11-188): the step right after the hot path
// (main.cc:160).  Own writers for the three formats the reference offers; same file names (<name>.ppm/.bmp/.hdr),
// same tone curve for the 8-bit formats (gamma_encoding, film.h:24), top row first in the film, and the format
// rules applied properly where the reference's writers are off:
//   * PPM "P3" samples are written as decimal numbers (the reference streams uint8_t, i.e. raw characters);
//   * BMP rows are padded to 4 bytes and the padded rows are what gets written (the reference computes the padded
//     layout but writes width*3-byte rows, so widths with width*3 % 4 != 0 come out sheared);
//   * HDR pixels below 1e-32 are written as zero RGBE (the reference leaves them uninitialised).
#include "jetpbrt.h"

#include <algorithm>
#include <cstring>
#include <fstream>

namespace jetpbrt
{
namespace
{
// 8-bit pixels of the film: the device-encoded bytes when the integrator delivered them (FFilm::RequestDeviceLDR), else
// gamma_encoding (film.h:24) of the fp32 pixels on the host -- the two are byte-identical (tests)
std::vector<uint8_t> Pixels8(int w, int h, const std::vector<FColor>& px, const std::vector<uint8_t>& ldr8)
{
	if (ldr8.size() == (size_t)w * h * 3) return ldr8;
	std::vector<uint8_t> b((size_t)w * h * 3);
	for (size_t i = 0; i < px.size(); i++) { b[3 * i] = gamma_encoding(px[i].r); b[3 * i + 1] = gamma_encoding(px[i].g); b[3 * i + 2] = gamma_encoding(px[i].b); }
	return b;
}

bool WritePPM(const std::string& path, int w, int h, const std::vector<uint8_t>& p8)
{
	std::ofstream f(path, std::ios::binary | std::ios::out);
	if (!f.is_open()) return false;
	f << "P3\n" << w << " " << h << "\n255\n";
	for (size_t i = 0; i < (size_t)w * h; i++)
		f << (int)p8[3 * i] << "  " << (int)p8[3 * i + 1] << "  " << (int)p8[3 * i + 2] << "\n";
	return (bool)f;
}

void put16(std::vector<uint8_t>& b, size_t at, uint16_t v) { b[at] = (uint8_t)v; b[at + 1] = (uint8_t)(v >> 8); }
void put32(std::vector<uint8_t>& b, size_t at, uint32_t v) { for (int k = 0; k < 4; k++) b[at + k] = (uint8_t)(v >> (8 * k)); }

bool WriteBMP(const std::string& path, int w, int h, const std::vector<uint8_t>& p8)
{
	const size_t row = ((size_t)w * 3 + 3) & ~(size_t)3, body = row * h, head = 14 + 40;
	std::vector<uint8_t> b(head + body, 0);
	b[0] = 'B'; b[1] = 'M';
	put32(b, 2, (uint32_t)(head + body)); put32(b, 10, (uint32_t)head);
	put32(b, 14, 40); put32(b, 18, (uint32_t)w); put32(b, 22, (uint32_t)h); put16(b, 26, 1); put16(b, 28, 24);
	// biSizeImage stays 0, as the reference writes it (legal for BI_RGB): for widths whose rows need no padding the file is then
	// byte-identical to the reference's (tests/golden/film_io.npz)
	for (int y = 0; y < h; y++)                                  // BMP stores the bottom row first
	{
		uint8_t* line = &b[head + row * (size_t)(h - 1 - y)];
		for (int x = 0; x < w; x++)
		{
			const uint8_t* c = &p8[3 * ((size_t)y * w + x)];
			line[3 * x + 0] = c[2]; line[3 * x + 1] = c[1]; line[3 * x + 2] = c[0];                   // B G R
		}
	}
	std::ofstream f(path, std::ios::binary | std::ios::out);
	if (!f.is_open()) return false;
	f.write((const char*)b.data(), (std::streamsize)b.size());
	return (bool)f;
}

bool WriteHDR(const std::string& path, int w, int h, const std::vector<FColor>& px)
{
	std::ofstream f(path, std::ios::binary | std::ios::out);
	if (!f.is_open()) return false;
	f << "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y " << h << " +X " << w << "\n";
	for (size_t i = 0; i < px.size(); i++)
	{
		uint8_t rgbe[4] = { 0, 0, 0, 0 };
		const FColor& c = px[i];
		const float v = std::max(c.r, std::max(c.g, c.b));
		if (v >= 1e-32f)
		{
			int e; const float m = (float)(std::frexp(v, &e) * 256.f / v);   // v = mantissa * 2^e; channel byte = channel * 256 * mantissa / v
			rgbe[0] = (uint8_t)(c.r * m); rgbe[1] = (uint8_t)(c.g * m); rgbe[2] = (uint8_t)(c.b * m); rgbe[3] = (uint8_t)(e + 128);
		}
		f.write((const char*)rgbe, 4);
	}
	return (bool)f;
}
}

bool FFilm::SaveAsImage(const std::string& filename, EImageType imgType) const
{
	switch (imgType)
	{
	case EImageType::PPM: return WritePPM(filename + ".ppm", width, height, Pixels8(width, height, pixels, ldr8));
	case EImageType::BMP: return WriteBMP(filename + ".bmp", width, height, Pixels8(width, height, pixels, ldr8));
	case EImageType::HDR: return floatValid ? WriteHDR(filename + ".hdr", width, height, pixels) : false;   // an LDR-only render left no fp32 pixels to write
	}
	return false;
}

} // namespace jetpbrt
