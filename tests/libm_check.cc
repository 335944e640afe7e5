// tests/libm_check.cc -- jp_libm.h (the device's transcription of glibc's logf / expf / powf / acosf / atanf / tanf) against the
// running libm, bit for bit.  Test infrastructure: compiled and run by tests/test_libm_exact.py.
//   libm_check N      -> one line per function: mismatches of the FMA build / of the plain build / arguments
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../jet-pbrt_amd/csrc/jp_libm.h"
using namespace jp::lm;
static uint32_t st = 0x9e3779b9u;
static inline uint32_t rnd() { st = st * 1664525u + 1013904223u; return st; }
static inline float u01() { return (float)(rnd() >> 8) * (1.0f / 16777216.0f); }
static inline bool same(float a, float b) { uint32_t x, y; memcpy(&x, &a, 4); memcpy(&y, &b, 4); return x == y || (a != a && b != b); }
int main(int argc, char** argv)
{
	const long N = argc > 1 ? atol(argv[1]) : 1000000;
	long bad[8][2] = { { 0 } };
	for (long i = 0; i < N; i++)
	{
		// arguments: the call sites' ranges plus raw bit patterns (every exponent, both signs, specials)
		const int kind = i % 4;
		float x, y;
		if (kind == 0) { uint32_t b = rnd(); memcpy(&x, &b, 4); uint32_t c = rnd(); memcpy(&y, &c, 4); }
		else if (kind == 1) { x = u01(); y = u01() * 8.f; }
		else if (kind == 2) { x = (u01() - 0.5f) * 250.f; y = (u01() - 0.5f) * 64.f; }
		else { x = u01() * 1e-3f; y = 1.f / (u01() * 100.f + 1.f); }
		{ const float h = expf(x); if (!same(h, expf_libm<true>(x))) bad[0][0]++; if (!same(h, expf_libm<false>(x))) bad[0][1]++; }
		{ const float h = logf(x); if (!same(h, logf_libm<true>(x))) bad[1][0]++; if (!same(h, logf_libm<false>(x))) bad[1][1]++; }
		{ const float h = powf(x, y); if (!same(h, powf_libm<true>(x, y))) { if (bad[2][0]++ < 3 && argc > 2) printf("powf(%a, %a) = %a, ours %a\n", x, y, h, powf_libm<true>(x, y)); } if (!same(h, powf_libm<false>(x, y))) bad[2][1]++; }
		{ const float a = kind == 0 ? x : x * 2.f - 1.f; const float h = acosf(a); if (!same(h, acosf_libm(a))) { if (bad[3][0]++ < 3 && argc > 2) printf("acosf(%a) = %a, ours %a\n", a, h, acosf_libm(a)); } }
		{ const float h = atanf(x); if (!same(h, atanf_libm(x))) { if (bad[4][0]++ < 3 && argc > 2) printf("atanf(%a) = %a, ours %a\n", x, h, atanf_libm(x)); } }
		{ bool ok; const float t = tanf_libm(x, &ok); if (ok) { const float h = tanf(x); if (!same(h, t)) { if (bad[5][0]++ < 3 && argc > 2) printf("tanf(%a) = %a, ours %a\n", x, h, t); } if (!same(h, tanf_libm(x, &ok))) bad[5][1]++; } }
		{ const float a = x * 8.f; bool ok; const float t = tanf_libm(a, &ok); if (ok) { const float h = tanf(a); if (!same(h, t)) { if (bad[6][0]++ < 3 && argc > 2) printf("tanf(%a) = %a, ours %a\n", a, h, t); } if (!same(h, tanf_libm(a, &ok))) bad[6][1]++; } }
	}
	const char* names[7] = { "expf", "logf", "powf", "acosf", "atanf", "tanf", "tanf8" };
	for (int f = 0; f < 7; f++) printf("%s %ld %ld %ld\n", names[f], bad[f][0], bad[f][1], N);
	return 0;
}
