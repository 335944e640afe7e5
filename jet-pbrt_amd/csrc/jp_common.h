// jet-pbrt_amd/csrc/jp_common.h -- what the wavefront kernels share: queue layout, per-launch constants, the sampler draw,
// the pixel enumeration and the wave-level LDS take.  Included by jp_kernels.hip and jp_path.h.
#pragma once
#include "jp_shading.h"

using namespace jp;

extern __shared__ float4 s_dyn[];      // dynamic LDS of every kernel, 16-byte aligned base

#define JP_SHADE_CLASSES 6             // material classes of k_shade's sort: none / matte / mirror / glass / plastic / metal

// ---------------------------------------------------------------------------------------------------------------------
// device structures
// ---------------------------------------------------------------------------------------------------------------------
struct DevCounters
{
	unsigned int n_queue[2];                   // totals (host-side drain check for null-material scenes, statistics)
	unsigned int n_shadow;
	unsigned int pad;
	unsigned long long closest, closest_hit, shadow, shadow_occ;
	unsigned long long cert_fallback;          // Walker<6>: rays walked again the reference's way
};

struct Queues
{
	float4 *ray_o[2], *ray_d[2], *beta[2];     // ping-pong ray queues: (o, slot) (d, flags) (beta, key)
	float2 *hit;                               // (t, device prim index or -1) per queued ray
	float4 *lacc;                              // per slot: radiance of the path so far
	float4 *sh_o;                              // per shadow entry: (origin, slot | count << RenderConst::slot_bits)
	float4 *sh_d, *sh_c;                       // plane k at [k * cap + q]: (dir, tmax), (contribution, visible flag)
	unsigned int *blk_q[2], *blk_sh;           // per-block fill of the regions
	unsigned int cap;                          // G * R entries per queue array
	unsigned int R;                            // region capacity (multiple of JP_BLOCK); block b owns [b*R, (b+1)*R)
};

struct RenderConst
{
	int width, height, spp, max_depth;
	unsigned int seed;
	int band_rows, shard_index, shard_count;
	int npix;            // pixels of this shard
	int local_rows;
	int s0, sbatch;      // first sample index and sample count of this batch
	int n_planes;        // shadow ray planes (lights that can emit)
	int tiled;           // pixel enumeration: 16 x 4 tiles (1) or row-major (0)
	int compact;         // k_raygen: a region takes consecutive (pixel block, sample) chunks (1) or an even sample of the image (0)
	int slot_bits;       // a shadow entry's header word: slot in the low slot_bits bits, ray count above (27 + 5, or 24 + 8 for scenes with more than 31 emitting lights)
	int sampler_debug;   // JP_SAMPLER_DEBUG: every draw is 0.5
	int class_mask;      // material classes present in the scene (bit 0: none / null material, bit 1 + JP_MAT_*), k_shade<kSort>
	int lane_index, lane_count, lane_rows;   // stream lanes: this launch owns the lane_rows-row groups g of the shard's rows with g % lane_count == lane_index
};

// one draw: the counter stream, or FDebugSampler's constant (sampler.h:109-127: GetFloat 0.5, GetFloat2 (0.5, 0.5), camera sample pixel + 0.5)
__device__ __forceinline__ float rngf(const RenderConst& rc, uint32_t key, uint32_t dim) { return rc.sampler_debug ? 0.5f : jp_rng_float(key, dim); }

#define FLAG_BOUNCE(f) ((f) & 0xff)
#define FLAG_SPEC(f)   (((f) >> 8) & 1)
#define FLAG_DIM(f)    (((unsigned)(f)) >> 16)
#define MK_FLAGS(bounce, spec, dim) (((bounce) & 0xff) | ((spec) ? 0x100 : 0) | ((int)(dim) << 16))

// pixel index of this shard -> film coordinates.  With rc.tiled (large scenes, image a whole number of 16 x 4 tiles) pixels
// are enumerated tile by tile, so the 64 lanes of a wave start as a 16 x 4 patch of the image: camera rays and their first
// shadow rays take nearly the same way through the scene (fewer divergent leaf visits, better cache reuse).
__device__ __forceinline__ void pixel_of(const RenderConst& rc, int pix, int& x, int& y)
{
	int lx, r;
	if (rc.tiled)
	{
		const int t = pix >> 6, i = pix & 63, tpr = rc.width >> 4;
		const int ty = t / tpr, tx = t - ty * tpr;
		lx = (tx << 4) + (i & 15); r = (ty << 2) + (i >> 4);
	}
	else { r = pix / rc.width; lx = pix - r * rc.width; }
	x = lx;
	if (rc.lane_count > 1) { const int m = r / rc.lane_rows; r = (rc.lane_index + m * rc.lane_count) * rc.lane_rows + (r - m * rc.lane_rows); }   // lane row -> shard row
	const int j = r / rc.band_rows;
	y = (rc.shard_index + j * rc.shard_count) * rc.band_rows + (r - j * rc.band_rows);
}

// wave-level take from an LDS counter: lane 0 adds the wave-uniform n, every lane gets the old value.  All 64 lanes must be active.
__device__ __forceinline__ unsigned int wave_take(unsigned int* ctr, unsigned int n)
{
	unsigned int v = 0;
	if ((threadIdx.x & 63) == 0) v = atomicAdd(ctr, n);
	return (unsigned int)__builtin_amdgcn_readfirstlane((int)v);
}

