// jet-pbrt_amd/csrc/jp_kernels.hip -- the wavefront path tracer: hand-written gfx950 HIP kernels plus the C ABI
// of include/jetpbrt_amd.h.  Replaces the per-pixel / per-sample loop of the reference
// (FIntegrator::Render integrator.cc:35-80 -> DoRender :82-111 -> FPathIntegratorIteration::Li :316-403).
//
// One batch = all pixels of this GPU's bands x S samples = P path slots (slot = s_local * NPIX + pixel).
//   k_raygen   camera samples -> ray queue                         (sampler.h:148-155, camera.h:52-58)
//   per bounce:
//   k_extend   closest hit per queued ray: BVH traversal with an LDS stack, scene in LDS when it fits; tiny scenes use a
//              flat wide node tested wave-uniformly
//   k_shade    emission, material closure, NEE light samples -> shadow rays, BSDF sample, Russian roulette,
//              surviving paths compacted (wave ballots, room taken from an LDS counter per wave) into the next ray queue
//   k_shadow   any-hit traversal per shadow entry, visible contributions added to the path's radiance in light order
//   k_resolve  per pixel: sequential fp32 sum over the batch's samples in index order (integrator.cc:102-105)
// Round 2: k_shade partitions its workgroup's region by material class first (wave ballots + one block scan of LDS counters) and
// shades it in 64-path chunks the waves take from an LDS counter, expensive classes first, no barrier in between;
// large scenes trace through k_extend_persist / k_shadow_persist (resumable Walker<mode> steps, idle lanes refilled from the
// region, per-iteration vote); k_extend_sort / k_shadow_sort (opt-in) partition rays by expected work; k_tonemap8 delivers the
// film as 8-bit gamma-encoded RGB; k_bsdf evaluates any BSDF class of bsdf.h by value (jp_xbsdf.h).
// Queues are SoA float4 arrays in HBM cut into one REGION per workgroup: block b reads region b of the input queue
// and appends to region b of the output queues with a running offset, so compaction needs no global atomic and the
// layout is deterministic.  Launches are asynchronous on one stream, no host round trip inside a batch.
#include "jp_common.h"
#include "jp_xbsdf.h"

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>
#include <cmath>
#include <mutex>
#include <thread>
#include "jp_lbvh.h"
#include "jp_ploc.h"

// ---------------------------------------------------------------------------------------------------------------------
// Wavefront-level sort: block-wide STABLE partition of a tile of kRPT x 256 queue entries by a small class key, with wave
// ballots + popcount prefixes and one block prefix over LDS counters (no atomics: the order is deterministic).
//   key[r]   class of the entry at tile position r * 256 + tid (>= kClasses: position beyond the tile, not placed)
//   s_cnt    kClasses * kRPT * 4 counters: entries of class c in (pass r, wave w), class-major -> after the scan their bases
//   s_idx    s_idx[sorted position] = tile position
// Three barriers; every thread of the block must call it.  Classes absent from class_mask (uniform) cost one store.
// Used by k_shade (material class) and by the sorted traversal kernels (expected traversal work).
// ---------------------------------------------------------------------------------------------------------------------
template <int kRPT, int kClasses>
__device__ __forceinline__ void tile_partition(const unsigned int (&key)[kRPT], unsigned int class_mask, unsigned int* s_cnt, unsigned short* s_idx)
{
	constexpr int kSeg = kRPT * (JP_BLOCK / 64), kN = kClasses * kSeg;
	static_assert(kN <= 128, "tile_partition: one wave scans two counters per lane");
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const unsigned long long lt = (1ull << lane) - 1ull;
	unsigned int pre[kRPT];
	#pragma unroll
	for (int r = 0; r < kRPT; r++)
	{
		pre[r] = 0;
		#pragma unroll
		for (int c = 0; c < kClasses; c++)
		{
			if (!((class_mask >> c) & 1u)) { if (lane == 0) s_cnt[c * kSeg + r * (JP_BLOCK / 64) + wave] = 0; continue; }
			const unsigned long long m = __ballot(key[r] == (unsigned int)c);
			if (key[r] == (unsigned int)c) pre[r] = (unsigned int)__popcll(m & lt);
			if (lane == 0) s_cnt[c * kSeg + r * (JP_BLOCK / 64) + wave] = (unsigned int)__popcll(m);
		}
	}
	__syncthreads();
	if (threadIdx.x < 64)
	{   // exclusive scan of the kN counters by one wave, two per lane
		const int a = threadIdx.x, b2 = threadIdx.x + 64;
		const unsigned int va = a < kN ? s_cnt[a] : 0u, vb = b2 < kN ? s_cnt[b2] : 0u;
		unsigned int ia = va, ib = vb;
		#pragma unroll
		for (int off = 1; off < 64; off <<= 1)
		{
			const unsigned int ta = __shfl_up(ia, off), tb = __shfl_up(ib, off);
			if (a >= off) { ia += ta; ib += tb; }
		}
		const unsigned int tot_a = __shfl(ia, 63);
		if (a < kN) s_cnt[a] = ia - va;
		if (b2 < kN) s_cnt[b2] = tot_a + ib - vb;
	}
	__syncthreads();
	#pragma unroll
	for (int r = 0; r < kRPT; r++)
		if (key[r] < (unsigned int)kClasses) s_idx[s_cnt[key[r] * kSeg + r * (JP_BLOCK / 64) + wave] + pre[r]] = (unsigned short)(r * JP_BLOCK + threadIdx.x);
	__syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------------
// k_raygen: FSampler::GetCameraSample (sampler.h:148-155) + FCamera::GenerateRay (camera.h:52-58).
// Block b takes the 256-slot chunks b, b+G, b+2G, ... (an even sample of the image, so every region ages alike) and
// writes them contiguously into region b of queue 0.
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(JP_BLOCK) k_raygen(SceneView sc, Queues q, RenderConst rc, DevCounters* cnt)
{
	const unsigned int total = (unsigned int)rc.npix * rc.sbatch;
	const unsigned int nchunks = (total + JP_BLOCK - 1) / JP_BLOCK;
	const unsigned int G = gridDim.x, b = blockIdx.x;
	if (b == 0 && threadIdx.x == 0) { cnt->n_queue[0] = total; cnt->n_queue[1] = 0; cnt->n_shadow = 0; }
	unsigned int filled = 0;
	// rc.compact (large scenes, round 3): region b takes CONSECUTIVE chunks of the list ordered by (256-pixel block, sample): a region is a
	// few pixel blocks with all their samples, and the ~2000 workgroups in flight cover one part of the image, so the nodes and
	// primitives their rays fetch are the same few MB (the L2 of an XCD holds 4 MB of a 27 MB scene).  Otherwise (scenes in LDS) the
	// chunks b, b + G, ...: an even sample of the image per region, regions age alike.
	const unsigned int K = (nchunks + G - 1) / G, npb = (unsigned int)rc.npix / JP_BLOCK;
	for (unsigned int k = 0, j0 = 0; k < K; k++, j0 += JP_BLOCK)
	{
		unsigned int c;
		if (rc.compact) { const unsigned int qi = b * K + k; if (qi >= nchunks) break; const unsigned int pb = qi / (unsigned int)rc.sbatch, sl = qi - pb * (unsigned int)rc.sbatch; c = sl * npb + pb; }
		else { c = b + k * G; if (c >= nchunks) break; }
		const unsigned int slot = c * JP_BLOCK + threadIdx.x;
		if (slot < total)
		{
			const unsigned int i = b * q.R + j0 + threadIdx.x;
			const int pix = slot % rc.npix, s = rc.s0 + slot / rc.npix;
			int x, y; pixel_of(rc, pix, x, y);
			const uint32_t key = jp_rng_key(rc.seed, (uint32_t)x, (uint32_t)y, (uint32_t)s);
			const float fx = (float)x + rngf(rc, key, 0), fy = (float)y + rngf(rc, key, 1);
			const V3 front = mk(sc.cam.front[0], sc.cam.front[1], sc.cam.front[2]);
			const V3 right = mk(sc.cam.right[0], sc.cam.right[1], sc.cam.right[2]);
			const V3 up = mk(sc.cam.up[0], sc.cam.up[1], sc.cam.up[2]);
			V3 dir = front + right * (fx / sc.cam.res_x - 0.5f) + up * (0.5f - fy / sc.cam.res_y);
			dir = normalize(dir);
			q.ray_o[0][i] = make_float4(sc.cam.pos[0], sc.cam.pos[1], sc.cam.pos[2], __int_as_float((int)slot));
			q.ray_d[0][i] = make_float4(dir.x, dir.y, dir.z, __int_as_float(MK_FLAGS(0, 0, 2)));
			q.beta[0][i] = make_float4(1.f, 1.f, 1.f, __int_as_float((int)key));
			q.lacc[slot] = make_float4(0.f, 0.f, 0.f, 0.f);
		}
		const unsigned int left = total - c * JP_BLOCK;
		filled += left < JP_BLOCK ? left : JP_BLOCK;
	}
	if (threadIdx.x == 0) q.blk_q[0][b] = filled;
}

// ---------------------------------------------------------------------------------------------------------------------
// LDS layout of the traversal kernels: [stack: depth * 256 ints][nodes][prims]
// ---------------------------------------------------------------------------------------------------------------------

// kMode 0: BVH and primitives in global memory (L2 / Infinity Cache resident), one stack plane in LDS
// kMode 1: BVH + primitives staged into LDS next to the stack
// kMode 2: tiny scene: flat leaf list (uniform loads from global), primitives in LDS, no stack
// kMode 3: large scene: 8-wide quantised BVH in global memory, (group, hits) stack pairs in LDS
// kMode 5: reference semantics: the caller's tree node for node, unordered, the reference's box test (traverse_ref)
// -DJP_SHADE_TIMING (diagnostic builds only, tools/shade_timing.py): every wave of k_shade adds the shader-clock cycles it spends
// in each section of the kernel to g_shade_t; jp_dbg_shade_timing reads and clears the sums.
// -DJP_TRAV_TIMING: the same for k_extend<2> / k_shadow<2> (sections listed in tools/shade_timing.py).
#if defined(JP_SHADE_TIMING) || defined(JP_TRAV_TIMING)
__device__ unsigned long long g_shade_t[16];
#define JP_TSX(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); t_acc[i] += t_ - t_last; t_last = t_; } while (0)
#endif
#ifdef JP_SHADE_TIMING
#define JP_TS(i) JP_TSX(i)
#else
#define JP_TS(i) do { } while (0)
#endif
#ifdef JP_TRAV_TIMING
#define JP_TT(i) JP_TSX(i)
#else
#define JP_TT(i) do { } while (0)
#endif
template <int kMode>
struct SceneAccess;
template <> struct SceneAccess<0>
{
	const float4 *nodes, *prims; int* stack; int depth;
	__device__ __forceinline__ SceneAccess(const SceneView& sc, int depth_) : nodes(sc.nodes), prims(sc.prims), stack((int*)s_dyn + threadIdx.x), depth(depth_) {}
	template <bool kAnyHit> __device__ __forceinline__ int trace(const SceneView&, V3 o, V3 d, float tmin, float& tmax) const
	{ return traverse<kAnyHit, 4>(nodes, prims, o, d, tmin, tmax, stack); }
};
template <> struct SceneAccess<1>
{
	float4 *nodes, *prims; int* stack; int depth;
	__device__ __forceinline__ SceneAccess(const SceneView& sc, int depth_) : depth(depth_)
	{
		stack = (int*)s_dyn + threadIdx.x;
		nodes = s_dyn + (depth * JP_BLOCK) / 4;
		prims = nodes + 5 * sc.n_nodes;              // 80-byte record stride in LDS (bank spreading), 64 bytes used
		for (int i = threadIdx.x; i < 4 * sc.n_nodes; i += JP_BLOCK) nodes[5 * (i >> 2) + (i & 3)] = sc.nodes[i];
		for (int i = threadIdx.x; i < 4 * sc.n_prims; i += JP_BLOCK) prims[5 * (i >> 2) + (i & 3)] = sc.prims[i];
		__syncthreads();
	}
	template <bool kAnyHit> __device__ __forceinline__ int trace(const SceneView&, V3 o, V3 d, float tmin, float& tmax) const
	{ return traverse<kAnyHit, 5>(nodes, prims, o, d, tmin, tmax, stack); }
};
template <> struct SceneAccess<2>
{
	float4* prims;
	__device__ __forceinline__ SceneAccess(const SceneView& sc, int)
	{
		prims = s_dyn;
		for (int i = threadIdx.x; i < 4 * sc.n_prims; i += JP_BLOCK) prims[5 * (i >> 2) + (i & 3)] = sc.prims[i];
		__syncthreads();
	}
	template <bool kAnyHit> __device__ __forceinline__ int trace(const SceneView& sc, V3 o, V3 d, float tmin, float& tmax) const
	{ return traverse_flat<kAnyHit, 5>(sc.flat, sc.n_flat, sc.n_prims, prims, o, d, tmin, tmax); }
};

template <> struct SceneAccess<3>
{
	const float4* prims; unsigned int* stack;
	__device__ __forceinline__ SceneAccess(const SceneView& sc, int) : prims(sc.prims), stack((unsigned int*)s_dyn + threadIdx.x) {}
	template <bool kAnyHit> __device__ __forceinline__ int trace(const SceneView& sc, V3 o, V3 d, float tmin, float& tmax) const
	{ return traverse_wide<kAnyHit>(sc.wide, prims, o, d, tmin, tmax, stack); }
};

template <> struct SceneAccess<4>
{   // the 4-wide quantised tree, one ray per lane to the end (k_trace); the whole stack in LDS
	WalkStack stack;
	__device__ __forceinline__ SceneAccess(const SceneView&, int depth) { stack.lds = (int*)s_dyn + threadIdx.x; stack.spill = nullptr; stack.cap = depth - 1; stack.stride = 0; }   // (the last word: Walker<4>'s dump slot)
	template <bool kAnyHit> __device__ __forceinline__ int trace(const SceneView& sc, V3 o, V3 d, float tmin, float& tmax) const
	{ return walk_ray<4, kAnyHit>(sc, o, d, tmin, tmax, stack); }
};

template <> struct SceneAccess<6>
{   // reference semantics, certified walk (Walker<6>) with the verbatim walk behind it for the rays it cannot certify (k_trace)
	WalkStack stack;
	__device__ __forceinline__ SceneAccess(const SceneView&, int depth) { stack.lds = (int*)s_dyn + threadIdx.x; stack.spill = nullptr; stack.cap = depth - 1; stack.stride = 0; }
	template <bool kAnyHit> __device__ __forceinline__ int trace(const SceneView& sc, V3 o, V3 d, float tmin, float& tmax) const
	{ return walk_ray<6, kAnyHit>(sc, o, d, tmin, tmax, stack); }
};

template <> struct SceneAccess<5>
{
	const float4 *nodes, *prims; int* stack;
	__device__ __forceinline__ SceneAccess(const SceneView& sc, int) : nodes(sc.nodes), prims(sc.prims), stack((int*)s_dyn + threadIdx.x) {}
	template <bool kAnyHit> __device__ __forceinline__ int trace(const SceneView&, V3 o, V3 d, float tmin, float& tmax) const
	{ return traverse_ref<kAnyHit>(nodes, prims, o, d, tmin, tmax, stack); }
};

// ---------------------------------------------------------------------------------------------------------------------
// k_extend: FScene::Intersect (scene.cc:25-33) for every ray of this block's region
// ---------------------------------------------------------------------------------------------------------------------
template <int kMode>
__global__ void __launch_bounds__(JP_BLOCK) k_extend(SceneView sc, Queues q, int cur, int depth, DevCounters* cnt)
{
	SceneAccess<kMode> acc(sc, depth);
	const unsigned int b = blockIdx.x, n = q.blk_q[cur][b], rbase = b * q.R;
	if (b == 0 && threadIdx.x == 0) { cnt->closest += cnt->n_queue[cur]; cnt->n_queue[cur ^ 1] = 0; cnt->n_shadow = 0; }
	unsigned int h = 0;
	// software prefetch: the next iteration's ray is requested before this iteration's traversal
	float4 ro = make_float4(0, 0, 0, 0), rd = make_float4(0, 0, 1, 0);
	if (threadIdx.x < n) { ro = q.ray_o[cur][rbase + threadIdx.x]; rd = q.ray_d[cur][rbase + threadIdx.x]; }
	for (unsigned int j = threadIdx.x; j < n; j += JP_BLOCK)
	{
		const unsigned int i = rbase + j;
		const float4 co = ro, cd = rd;
		if (j + JP_BLOCK < n) { ro = q.ray_o[cur][i + JP_BLOCK]; rd = q.ray_d[cur][i + JP_BLOCK]; }
		float tmax = JP_INF;                                         // FRay defaults geometry.h:399: min_t 0.001, max_t infinity
		const int hit = acc.template trace<false>(sc, xyz(co), xyz(cd), 0.001f, tmax);
		q.hit[i] = make_float2(tmax, __int_as_float(hit));
		h += hit >= 0 ? 1u : 0u;
	}
	for (int off = 32; off > 0; off >>= 1) h += __shfl_down(h, off);
	if ((threadIdx.x & 63) == 0 && h) atomicAdd(&cnt->closest_hit, (unsigned long long)h);
}

// ---------------------------------------------------------------------------------------------------------------------
// k_shade: the body of FPathIntegratorIteration::Li after the intersection (integrator.cc:328-399)
// ---------------------------------------------------------------------------------------------------------------------
// Table staging: lights + materials (kTab) and, for scenes whose primitives fit, primitive records + meta (kPrims)
// are copied into LDS once per block, so the dependent lookups of a shading event (hit -> primitive -> material /
// light) are LDS reads instead of a chain of global loads.
// kStage (<= 4 emitting lights): the NEE rays of a path are staged in LDS and a shadow entry is allocated only when at
// least one ray survived the rejections of integrator.cc:362-367, so k_shadow never meets an empty entry.
// kSort ("material sort"): the paths of the workgroup's region (<= JP_SHADE_TILE) are partitioned by the material class of the
// primitive they hit (none / matte / mirror / glass / plastic / metal) before they are shaded, with wave ballots + a block scan over
// LDS counters, and the region is then shaded in that order, expensive classes first: a wave holds paths of ONE class except at class boundaries, so the microfacet code of
// bsdf.cc / microfacet.cc runs with full waves on the paths that need it instead of with 10-15 % of the lanes in every wave
// (measured on the reference's Cornell scene: k_shade 3124 -> wave64 instructions per 64 paths at lane utilisation 0.42 before).
// The partition is stable, so the big class still reads its records almost in queue order.  Every path computes exactly what it
// computed before; only the order inside this block's output regions changes.
#ifndef JP_SHADE_TILE
#define JP_SHADE_TILE 8192
#endif
template <bool kTab, bool kPrims, bool kStage, bool kSort>
__global__ void __launch_bounds__(JP_BLOCK) k_shade(SceneView sc, Queues q, RenderConst rc, int cur, DevCounters* cnt)
{
	static_assert(kTab || !kPrims, "k_shade: primitive records in LDS only together with the tables");
	constexpr int kWaves = JP_BLOCK / 64, kMaxSeg = (JP_SHADE_TILE / JP_BLOCK) * kWaves;  // (pass, wave) segments of a tile, in queue order
	static_assert(JP_SHADE_TILE % JP_BLOCK == 0 && JP_SHADE_TILE <= 65536, "k_shade: tile positions are 16-bit");
	__shared__ unsigned short s_idx[kSort ? JP_SHADE_TILE : 1];
	__shared__ unsigned char s_key[kSort ? JP_SHADE_TILE : 1];
	__shared__ unsigned int s_cnt[kSort ? JP_SHADE_CLASSES * kMaxSeg : 1];
	__shared__ unsigned int s_wsum[kWaves];
	__shared__ unsigned int s_ctr[3];      // [0] next 64-path chunk of the tile; [1], [2] fill of this block's ray / shadow output regions
	const unsigned int lane = threadIdx.x & 63u;
	// LDS tables in the order of sc.shade_tab: lights | mats | mat_type (padded to 16 B) | prims | meta | frames
	const int n_tab = 2 * sc.n_lights + 4 * sc.n_mats + (sc.n_mats + 3) / 4, n_tab_all = n_tab + (kPrims ? 8 * sc.n_prims : 0);
	float4* s_lights = s_dyn;
	float4* s_mats = s_lights + 2 * sc.n_lights;
	int* s_mtype = (int*)(s_mats + 4 * sc.n_mats);
	float4* s_prims = s_dyn + n_tab;
	int4* s_meta = (int4*)(s_prims + 4 * sc.n_prims);
	const float4* s_frames = s_prims + 5 * sc.n_prims;           // kPrims: the shading frame of every flat primitive, 3 x float4 (n, s, t)
	// NEE staging [(2k, 2k+1) * 256 + tid], behind the tables (an index into s_dyn, not a cast through an integer: the pointer keeps
	// its LDS address space, so the staging is ds_write / ds_read -- as flat accesses its reads sat behind `s_waitcnt vmcnt(0)`,
	// i.e. behind the acknowledgement of every store issued before them)
	float4* s_stage = s_dyn + n_tab_all + threadIdx.x;
	const unsigned int b = blockIdx.x, n = q.blk_q[cur][b];
	const int nxt = cur ^ 1;
#ifdef JP_SHADE_TIMING
	unsigned long long t_acc[10] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 }, t_last = __builtin_amdgcn_s_memtime();
#endif
	if (kTab)
	{   // (before the test of n: the table loads and the load of n are in flight together)
		// one array, one sweep, two loads in flight per thread: a single round trip to memory as a rule (five dependent copy loops
		// over five arrays were 13 % of a wave's lifetime on the Cornell box)
		for (int i = threadIdx.x; i < n_tab_all; i += 2 * JP_BLOCK)
		{
			const bool two = i + JP_BLOCK < n_tab_all;
			const float4 a = sc.shade_tab[i], c2 = two ? sc.shade_tab[i + JP_BLOCK] : a;
			s_dyn[i] = a; if (two) s_dyn[i + JP_BLOCK] = c2;
		}
	}
	if (n == 0) { if (threadIdx.x == 0) { q.blk_q[nxt][b] = 0; q.blk_sh[b] = 0; } return; }
	if (threadIdx.x == 0) { s_ctr[0] = 0; s_ctr[1] = 0; s_ctr[2] = 0; }
	__syncthreads();
	const float4* lights = kTab ? (const float4*)s_lights : sc.lights;
	const float4* mats = kTab ? (const float4*)s_mats : sc.mats;
	const int* mat_type = kTab ? (const int*)s_mtype : sc.mat_type;
	const float4* prims = kPrims ? (const float4*)s_prims : sc.prims;
	const int4* meta_t = kPrims ? (const int4*)s_meta : sc.meta;
	const unsigned int rbase = b * q.R;
	const unsigned int t0 = 0, count = n;                        // one tile: the host keeps R <= JP_SHADE_TILE for the sorted variant
	JP_TS(0);                                                     // [0] table staging
	if (kSort)
	{   // ---- stable partition of the tile's paths by material class: s_idx[sorted position] = position in the tile ----
		// Two sweeps over the tile (the whole region of the block as a rule: one partition and one closing barrier per launch):
		// count per (class, pass, wave) with wave ballots, one block-wide exclusive scan of the counters in class-major order,
		// then every entry goes to its class's base + its rank in its (pass, wave) segment.  Stable, no atomics.
		constexpr unsigned int seg = kMaxSeg, nctr = JP_SHADE_CLASSES * kMaxSeg;
		const unsigned int tid = threadIdx.x;
		const unsigned int plane = tid & 63u, wave = tid >> 6, npass = (count + JP_BLOCK - 1) / JP_BLOCK;
		const unsigned long long ltm = (1ull << plane) - 1ull;
		#pragma unroll 1
		for (unsigned int r0 = 0; r0 < npass; r0 += 8)
		{
			int pi[8];                                                   // eight hit records in flight per thread
			#pragma unroll
			for (int u = 0; u < 8; u++)
			{
				const unsigned int j = (r0 + u) * JP_BLOCK + tid;
				pi[u] = j < count ? __float_as_int(q.hit[rbase + t0 + j].y) : -2;
			}
			#pragma unroll
			for (int u = 0; u < 8; u++)
			{
				const unsigned int r = r0 + u;
				if (r >= npass) break;
				unsigned int key = JP_SHADE_CLASSES;                     // beyond the tile: no class
				if (pi[u] != -2)
				{
					int m = -1; if (pi[u] >= 0) m = meta_t[pi[u]].y;
					key = m >= 0 ? 1u + (unsigned int)mat_type[m] : 0u;
				}
				s_key[r * JP_BLOCK + tid] = (unsigned char)key;
				unsigned int mine = 0;                                   // lane c keeps the wave's count of class c: one LDS write per pass
				#pragma unroll
				for (int c = 0; c < JP_SHADE_CLASSES; c++)
				{
					unsigned int nc = 0;
					if ((rc.class_mask >> c) & 1) nc = (unsigned int)__popcll(__ballot(key == (unsigned int)c));
					if (plane == (unsigned int)c) mine = nc;
				}
				if (plane < JP_SHADE_CLASSES) s_cnt[plane * seg + r * kWaves + wave] = mine;
			}
		}
		__syncthreads();
		{   // exclusive scan of the nctr <= 3 * JP_BLOCK counters: three per thread, wave scan, wave totals through LDS
			static_assert(JP_SHADE_CLASSES * kMaxSeg <= 3 * JP_BLOCK, "k_shade: three counters per thread");
			unsigned int v[3], t = 0;
			#pragma unroll
			for (int i = 0; i < 3; i++)
			{   // counters of passes beyond the tile's last one were not written: they count nothing
				const unsigned int at = 3 * tid + i;
				v[i] = (at < nctr && (at % seg) / kWaves < npass) ? s_cnt[at] : 0u; t += v[i];
			}
			unsigned int incl = t;
			#pragma unroll
			for (int off = 1; off < 64; off <<= 1) { const unsigned int o = __shfl_up(incl, off); if (plane >= (unsigned int)off) incl += o; }
			if (plane == 63) s_wsum[wave] = incl;
			__syncthreads();
			unsigned int base = incl - t;
			#pragma unroll
			for (int w = 0; w < kWaves; w++) if ((unsigned int)w < wave) base += s_wsum[w];
			#pragma unroll
			for (int i = 0; i < 3; i++) { const unsigned int at = 3 * tid + i; if (at < nctr) s_cnt[at] = base; base += v[i]; }
		}
		__syncthreads();
		#pragma unroll 1
		for (unsigned int r = 0; r < npass; r++)
		{
			const unsigned int key = s_key[r * JP_BLOCK + tid];
			unsigned int pre = 0;
			#pragma unroll
			for (int c = 0; c < JP_SHADE_CLASSES; c++)
			{
				if (!((rc.class_mask >> c) & 1)) continue;
				const unsigned long long m = __ballot(key == (unsigned int)c);
				if (key == (unsigned int)c) pre = (unsigned int)__popcll(m & ltm);
			}
			if (key < JP_SHADE_CLASSES) s_idx[s_cnt[key * seg + r * kWaves + wave] + pre] = (unsigned short)(r * JP_BLOCK + tid);
		}
		__syncthreads();
	}
	// The tile is shaded in 64-path chunks that the waves take from an LDS counter: a wave with expensive paths (a chunk of the
	// microfacet class) takes fewer chunks, and no wave waits for another before the end of the tile.  The records of the chunk
	// a wave takes next are fetched while it shades the current one (software prefetch).
	float4 ro_n = make_float4(0, 0, 0, 0), rd_n = ro_n, rb_n = ro_n; float2 h_n = make_float2(0, 0);
	// Chunks are taken from the END of the sorted tile: the expensive classes (plastic, metal) sort last, and taking them first
	// leaves the cheap chunks to even out the waves before the barrier at the end of the tile.
	JP_TS(1);                                                     // [1] partition
	const unsigned int nch = (count + 63u) >> 6;
	unsigned int tk = wave_take(&s_ctr[0], 1u);                  // wave-uniform
	unsigned int c0 = (nch - 1u - tk) << 6;                      // first tile position of the wave's chunk (meaningful while tk < nch)
	if (tk < nch && c0 + lane < count) { const unsigned int pos = c0 + lane, i0 = rbase + t0 + (kSort ? (unsigned int)s_idx[pos] : pos); ro_n = q.ray_o[cur][i0]; rd_n = q.ray_d[cur][i0]; rb_n = q.beta[cur][i0]; h_n = q.hit[i0]; }
	while (tk < nch)
	{
		const bool valid = c0 + lane < count;
#ifdef JP_SHADE_TIMING
		JP_TS(2);                                                 // [2] loop overhead
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		JP_TS(3);                                                 // [3] wait for the prefetched records (and for the stores before them)
#endif
#ifdef JP_SHADE_NO_PREFETCH
		if (valid) { const unsigned int pos = c0 + lane, i0 = rbase + t0 + (kSort ? (unsigned int)s_idx[pos] : pos); ro_n = q.ray_o[cur][i0]; rd_n = q.ray_d[cur][i0]; rb_n = q.beta[cur][i0]; h_n = q.hit[i0]; }
		const float4 ro = ro_n, rd = rd_n, rb = rb_n; const float2 h = h_n;
		tk = wave_take(&s_ctr[0], 1u); c0 = (nch - 1u - tk) << 6;
#else
		const float4 ro = ro_n, rd = rd_n, rb = rb_n; const float2 h = h_n;
		tk = wave_take(&s_ctr[0], 1u); c0 = (nch - 1u - tk) << 6;
		if (tk < nch && c0 + lane < count)
		{
			const unsigned int pos = c0 + lane;
			const unsigned int i1 = rbase + t0 + (kSort ? (unsigned int)s_idx[pos] : pos);
			ro_n = q.ray_o[cur][i1]; rd_n = q.ray_d[cur][i1]; rb_n = q.beta[cur][i1]; h_n = q.hit[i1];
		}
#endif
		bool shaded = false, wantNee = false, alive = false;
		V3 o = mk(0, 0, 0), d = mk(0, 0, 1), beta = mk(0, 0, 0), p = mk(0, 0, 0), N = mk(0, 0, 1);
		int slot = 0, bounce = 0; bool spec = false; unsigned int dim = 0; uint32_t key = 0;
		Closure c; c.kind = CL_LAMBERT; Frame fr; fr.s = fr.t = fr.n = mk(0, 0, 1);
		if (valid)
		{
			o = xyz(ro); d = xyz(rd); beta = xyz(rb);
			slot = __float_as_int(ro.w); key = (uint32_t)__float_as_int(rb.w);
			const int flags = __float_as_int(rd.w);
			bounce = FLAG_BOUNCE(flags); spec = FLAG_SPEC(flags); dim = FLAG_DIM(flags);
			const int pi = __float_as_int(h.y);
			const bool found = pi >= 0;
			int mat = -1, hitprim = 0; bool nflip = false, tabframe = kPrims;
			V3 Le = splat(0);
			if (found)
			{
				const float4 g3 = prims[4 * pi + 3];
				const int4 meta = meta_t[pi];
				const int type = __float_as_int(g3.w);
				p = o + h.x * d;                                                      // ray(distance) geometry.h:412-416
				if (type == JP_SHAPE_TRIANGLE) N = xyz(g3);
				else if (type == JP_SHAPE_RECTANGLE) { nflip = !(dot(xyz(g3), d) <= 0); N = nflip ? -xyz(g3) : xyz(g3); }   // shape.h:427
				else if (type == JP_SHAPE_DISK) N = xyz(prims[4 * pi + 1]);                           // shape.h:214
				else { const float4 g0 = prims[4 * pi]; N = normalize(p - xyz(g0)); tabframe = false; }   // shape.h:521
				hitprim = pi;
				mat = meta.y;
				if (meta.z >= 0 && (bounce == 0 || spec))                             // primitive.h:60-63, light.h:234-238
				{
					const V3 wo = -d;
					if (dot(N, wo) > 0.f) Le = xyz(lights[2 * meta.z]);
				}
			}
			else if (bounce == 0 || spec)                                             // integrator.cc:334-336, light.h:300-303
			{
				// L += beta * Le for each infinite light in order; folded on the host only when there is at most one
				for (int li = 0; li < sc.n_lights; li++)
				{
					const float4 l0 = lights[2 * li];
					if (__float_as_int(l0.w) == JP_LIGHT_ENVIRONMENT && !isblack(xyz(l0)))
					{
						float4 L = q.lacc[slot];
						V3 a = mk(L.x, L.y, L.z) + cmul(beta, xyz(l0));
						q.lacc[slot] = make_float4(a.x, a.y, a.z, 0.f);
					}
				}
			}
			if (!isblack(Le))
			{
				float4 L = q.lacc[slot];
				V3 a = mk(L.x, L.y, L.z) + cmul(beta, Le);                            // integrator.cc:331
				q.lacc[slot] = make_float4(a.x, a.y, a.z, 0.f);
			}
			if (found && bounce < rc.max_depth)                                       // integrator.cc:340-343
			{
				if (mat < 0) alive = true;                                            // integrator.cc:349-353: pass through, same bounce
				else
				{
					float up = 0.f;
					const int mtype = mat_type[mat];
					if (mtype == JP_MAT_PLASTIC) up = rngf(rc, key, dim++);   // material.cc:14
					make_closure(mats, mtype, mat, up, c);
#ifdef JP_DBG_SKIP_FRAME
					fr.n = N; fr.s = mk(N.y, N.z, N.x); fr.t = mk(N.z, N.x, N.y);
#else
					if (kPrims && tabframe)
					{   // FFrame(normal) geometry.h:345-349 from the table the host computed with the same operations in the same order;
						// for the far side of a rectangle n and t change sign and s does not (every product and quotient keeps its
						// magnitude; |n.x| > 0.99 picks the same helper axis)
						const float4 fn = s_frames[3 * hitprim], fs = s_frames[3 * hitprim + 1], ft = s_frames[3 * hitprim + 2];
						fr.n = nflip ? -xyz(fn) : xyz(fn); fr.s = xyz(fs); fr.t = nflip ? -xyz(ft) : xyz(ft);
					}
					else fr = frame_from_z(N);
#endif
					shaded = true;
					wantNee = !is_delta(c);
#ifdef JP_DBG_SKIP_NEE
					wantNee = false;
#endif
				}
			}
		}
		JP_TS(4);                                                 // [4] prefetch issue, hit decode, emission, closure, frame
		// ---- next-event estimation (integrator.cc:357-372) ----
		unsigned int qs = 0;
		if (!kStage)
		{
			const unsigned long long mn = __ballot(wantNee);
			if (mn) qs = rbase + wave_take(&s_ctr[2], (unsigned int)__popcll(mn)) + (unsigned int)__popcll(mn & ((1ull << lane) - 1ull));
		}
		V3 nd = d, nbeta = beta; int nbounce = bounce; bool nspec = spec;
		V3 wo = mk(0, 0, 1);
		int k = 0;
		if (shaded)
		{
			const V3 wo_w = -d;
			wo = to_local(fr, wo_w);
			closure_set_wo(c, wo);
			if (wantNee)
			{
				for (int li = 0; li < sc.n_lights; li++)
				{
					const unsigned int d0 = dim; dim += 2;                              // the two draws are consumed even when the sample is rejected
					const float4 lrad = lights[2 * li];
					if (isblack(xyz(lrad))) continue;                                   // Li would be black (integrator.cc:362): skip the evaluation, keep the draws
					const float ux = rngf(rc, key, d0), uy = rngf(rc, key, d0 + 1);
					LightSample ls = sample_li(sc, prims, lights, li, p, N, ux, uy);
					if (isblack(ls.Li) || ls.pdf == 0.f) continue;
					const V3 f = eval_local(c, wo, to_local(fr, ls.wi));               // FBSDF::Evalf bsdf.h:284-287
					if (isblack(f)) continue;
					// FScene::Occluded scene.h:36-47: dir and distance recomputed from the sampled position
					// (for an area light Normalize(target - position) is the very expression that produced ls.wi)
					const V3 sdir = __float_as_int(lrad.w) == JP_LIGHT_AREA ? ls.wi : normalize(ls.pos - p);
					const float dist = ls.dist >= 0.f ? ls.dist : len(p - ls.pos);
					const V3 contrib = cmul(cmul(beta, f), ls.Li) * absdot(ls.wi, N) / ls.pdf;   // integrator.cc:369
					if (k < rc.n_planes)
					{
						if (kStage)
						{
							s_stage[(2 * k) * JP_BLOCK] = make_float4(sdir.x, sdir.y, sdir.z, dist - 0.001f);
							s_stage[(2 * k + 1) * JP_BLOCK] = make_float4(contrib.x, contrib.y, contrib.z, 0.f);
						}
						else
						{
							q.sh_d[(size_t)k * q.cap + qs] = make_float4(sdir.x, sdir.y, sdir.z, dist - 0.001f);
							q.sh_c[(size_t)k * q.cap + qs] = make_float4(contrib.x, contrib.y, contrib.z, 0.f);
						}
						k++;
					}
				}
				if (!kStage) q.sh_o[qs] = make_float4(p.x, p.y, p.z, __int_as_float(slot | (k << rc.slot_bits)));
			}
		}
		JP_TS(5);                                                 // [5] next-event estimation
#ifdef JP_DBG_SKIP_SAMPLE
		if (false)
#else
		if (shaded)
#endif
		{
			// ---- BSDF sample (integrator.cc:375-379) ----
			const float ux = rngf(rc, key, dim), uy = rngf(rc, key, dim + 1); dim += 2;
			BsdfSample bs = sample_local(c, wo, ux, uy);
			bs.wi = to_world(fr, bs.wi);                                              // bsdf.h:295-301
			if (!(isblack(bs.f) || bs.pdf == 0.f))
			{
				nspec = (bs.flags & BS_SPECULAR) != 0;                                // integrator.cc:381
				if (bounce >= 3)                                                      // integrator.cc:383-393
				{
					const float qq = smax(0.05f, 1 - maxcomp(bs.f));
					const float ur = rngf(rc, key, dim++);
					if (!(ur < qq))
					{
						nbeta = cmul(beta, bs.f * absdot(bs.wi, N) / (bs.pdf * (1 - qq)));
						alive = true;
					}
				}
				else
				{
					nbeta = cmul(beta, bs.f * absdot(bs.wi, N) / bs.pdf);             // integrator.cc:397
					alive = true;
				}
				nd = bs.wi; nbounce = bounce + 1;
			}
		}
		// ---- compact survivors into the next ray queue (and, when staged, the shadow entries): the wave takes room for its survivors
		// from the block's fill counters (no barrier; the order of the waves' pieces in the region is whatever order they arrive in,
		// which no result depends on: every path owns its slot and its radiance sum) ----
		JP_TS(6);                                                 // [6] BSDF sample, Russian roulette
		unsigned int j = 0;
		const unsigned long long lt = (1ull << lane) - 1ull;
		const unsigned long long ma = __ballot(alive);
		if (kStage)
		{   // both fills in one LDS atomic: rays in the low half, shadow entries in the high half (a region holds <= 8192 < 2^16)
			const unsigned long long ms = __ballot(k > 0);
			if (ma | ms)
			{
				const unsigned int base = wave_take(&s_ctr[1], (unsigned int)__popcll(ma) | ((unsigned int)__popcll(ms) << 16));
				j = rbase + (base & 0xffffu) + (unsigned int)__popcll(ma & lt);
				qs = rbase + (base >> 16) + (unsigned int)__popcll(ms & lt);
			}
			if (k > 0)
			{
				q.sh_o[qs] = make_float4(p.x, p.y, p.z, __int_as_float(slot | (k << rc.slot_bits)));
				for (int kk = 0; kk < k; kk++)
				{
					q.sh_d[(size_t)kk * q.cap + qs] = s_stage[(2 * kk) * JP_BLOCK];
					q.sh_c[(size_t)kk * q.cap + qs] = s_stage[(2 * kk + 1) * JP_BLOCK];
				}
			}
		}
		else if (ma) j = rbase + wave_take(&s_ctr[1], (unsigned int)__popcll(ma)) + (unsigned int)__popcll(ma & lt);   // (shadow entries took their room before the light loop)
		if (alive)
		{
			q.ray_o[nxt][j] = make_float4(p.x, p.y, p.z, __int_as_float(slot));       // SpawnRay shape.h:61-64
			q.ray_d[nxt][j] = make_float4(nd.x, nd.y, nd.z, __int_as_float(MK_FLAGS(nbounce, nspec, dim)));
			q.beta[nxt][j] = make_float4(nbeta.x, nbeta.y, nbeta.z, __int_as_float((int)key));
		}
		JP_TS(7);                                                 // [7] room in the output regions, store issue
#ifdef JP_SHADE_TIMING
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		JP_TS(9);                                                 // [9] memory wait at the end of the chunk
#endif
	}
	JP_TS(2);
	__syncthreads();                                              // every wave has added its survivors to the fill counters
	JP_TS(8);                                                     // [8] closing barrier
#ifdef JP_SHADE_TIMING
	if (lane == 0) for (int i = 0; i < 10; i++) atomicAdd(&g_shade_t[i], t_acc[i]);
	if (lane == 0) atomicAdd(&g_shade_t[15], 1ull);
#endif
	if (threadIdx.x == 0)
	{
		const unsigned int run_q = kStage ? (s_ctr[1] & 0xffffu) : s_ctr[1], run_sh = kStage ? (s_ctr[1] >> 16) : s_ctr[2];
		q.blk_q[nxt][b] = run_q; q.blk_sh[b] = run_sh;
		if (run_q) atomicAdd(&cnt->n_queue[nxt], run_q);
		if (run_sh) atomicAdd(&cnt->n_shadow, run_sh);
	}
}

// ---------------------------------------------------------------------------------------------------------------------
// k_shadow: FScene::Occluded (scene.h:36-47) for the entry's rays in light order; L += contribution when visible
// (integrator.cc:367-370).  One lane owns a path's entry, so the path's radiance is summed in exactly the
// reference's order and the film is run-to-run deterministic (no float atomics).
// ---------------------------------------------------------------------------------------------------------------------
template <int kMode>
__global__ void __launch_bounds__(JP_BLOCK, 4) k_shadow(SceneView sc, Queues q, RenderConst rc, int depth, DevCounters* cnt)
{
#ifdef JP_TRAV_TIMING
	unsigned long long t_acc[10] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 }, t_last = __builtin_amdgcn_s_memtime();
#endif
	SceneAccess<kMode> acc(sc, depth);
	const unsigned int b = blockIdx.x, E = q.blk_sh[b], rbase = b * q.R;
	JP_TT(0);                                                      // [0] primitive records to LDS, region fill
	unsigned int rays = 0, occ = 0;
	// software prefetch: the next entry's header and first ray are requested before this entry is traced
	float4 so_n = make_float4(0, 0, 0, 0), sd_n = make_float4(0, 0, 1, 0);
	if (threadIdx.x < E) { so_n = q.sh_o[rbase + threadIdx.x]; sd_n = q.sh_d[rbase + threadIdx.x]; }
	for (unsigned int j = threadIdx.x; j < E; j += JP_BLOCK)
	{
		const unsigned int e = rbase + j;
#ifdef JP_TRAV_TIMING
		JP_TT(1);                                                  // [1] loop overhead
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		JP_TT(2);                                                  // [2] wait for the prefetched entry (and earlier stores)
#endif
		const float4 so = so_n; float4 sd = sd_n;
		if (j + JP_BLOCK < E) { so_n = q.sh_o[e + JP_BLOCK]; sd_n = q.sh_d[e + JP_BLOCK]; }
		const int packed = __float_as_int(so.w);
		const int slot = packed & ((1 << rc.slot_bits) - 1), n = (int)((unsigned int)packed >> rc.slot_bits);
		if (n == 0) continue;
		bool any = false;
		const float4 L = q.lacc[slot];                               // issued up front: its latency hides behind the traversal
		V3 a = mk(L.x, L.y, L.z);
		for (int k = 0; k < n; k++)
		{
			const float4 c4 = q.sh_c[(size_t)k * q.cap + e];          // needed only after the traversal
			float tmax = sd.w;
			const V3 dir = xyz(sd);
			if (k + 1 < n) sd = q.sh_d[(size_t)(k + 1) * q.cap + e];
			JP_TT(3);                                              // [3] load issue
#ifdef JP_TRAV_TIMING
			int hit;
			if constexpr (kMode == 2)
			{
				const unsigned int m = flat_boxes<false>(sc.flat, sc.n_flat, xyz(so), dir, 0.001f, tmax);
				JP_TT(4);                                          // [4] box phase
				hit = flat_prims<true, false, 5>(m, acc.prims, xyz(so), dir, 0.001f, tmax);
				JP_TT(5);                                          // [5] primitive phase
			}
			else hit = acc.template trace<true>(sc, xyz(so), dir, 0.001f, tmax);
#else
			const int hit = acc.template trace<true>(sc, xyz(so), dir, 0.001f, tmax);
#endif
			rays++;
			if (hit >= 0) occ++;
			else { a = a + xyz(c4); any = true; }
			JP_TT(6);                                              // [6] contribution (waits for its load)
		}
		if (any) q.lacc[slot] = make_float4(a.x, a.y, a.z, 0.f);
		JP_TT(7);                                                  // [7] radiance store
	}
#ifdef JP_TRAV_TIMING
	JP_TT(1);
	if ((threadIdx.x & 63) == 0) { for (int i = 0; i < 8; i++) atomicAdd(&g_shade_t[i], t_acc[i]); atomicAdd(&g_shade_t[15], 1ull); }
#endif
	for (int off = 32; off > 0; off >>= 1) { rays += __shfl_down(rays, off); occ += __shfl_down(occ, off); }
	if ((threadIdx.x & 63) == 0) { if (rays) atomicAdd(&cnt->shadow, (unsigned long long)rays); if (occ) atomicAdd(&cnt->shadow_occ, (unsigned long long)occ); }
}

#define JP_SORT_TILE 1024
#define JP_SORT_CLASSES 6
// expected-work class of a ray: tiny scenes (mode 2) -- the number of primitive tests it will pay (popcount of the box-phase
// mask, tools/flat_stats.py: mean 3.3, the unluckiest of 64 lanes ~10); other scenes -- the number of "cut" subtrees (the
// largest subtrees below the root, <= 16 boxes tested wave-uniformly like the flat list) its segment enters
__device__ __forceinline__ unsigned int work_class_flat(int pc) { return pc <= 1 ? 0u : pc == 2 ? 1u : pc <= 4 ? 2u : pc <= 6 ? 3u : pc <= 8 ? 4u : 5u; }
__device__ __forceinline__ unsigned int work_class_cut(int pc) { return pc < 5 ? (unsigned int)pc : 5u; }

// k_extend_sort: k_extend with the rays of each 1024-ray tile partitioned by expected work before they are traced, so that the
// 64 lanes of a wave finish together.  The hit record of every ray goes to the ray's own queue position: k_shade sees what it saw.
template <int kMode>
__global__ void __launch_bounds__(JP_BLOCK) k_extend_sort(SceneView sc, Queues q, int cur, int depth, DevCounters* cnt)
{
	constexpr int kRPT = JP_SORT_TILE / JP_BLOCK;
	__shared__ unsigned short s_idx[JP_SORT_TILE];
	__shared__ unsigned int s_cnt[JP_SORT_CLASSES * kRPT * (JP_BLOCK / 64)];
	__shared__ u64 s_mask[kMode == 2 ? JP_SORT_TILE : 1];           // mode 2: the box-phase result, so phase 1 runs once
	__shared__ unsigned int s_take;
	if (threadIdx.x == 0) s_take = 0;
	SceneAccess<kMode> acc(sc, depth);
	const unsigned int b = blockIdx.x, n = q.blk_q[cur][b], rbase = b * q.R;
	if (b == 0 && threadIdx.x == 0) { cnt->closest += cnt->n_queue[cur]; cnt->n_queue[cur ^ 1] = 0; cnt->n_shadow = 0; }
	const bool w64 = sc.n_prims > 32;
	unsigned int h = 0;
	for (unsigned int t0 = 0; t0 < n; t0 += JP_SORT_TILE)
	{
		const unsigned int count = n - t0 < (unsigned int)JP_SORT_TILE ? n - t0 : (unsigned int)JP_SORT_TILE;
		unsigned int key[kRPT];
		#pragma unroll
		for (int r = 0; r < kRPT; r++)
		{
			const unsigned int j = r * JP_BLOCK + threadIdx.x;
			key[r] = JP_SORT_CLASSES;
			if (j < count)
			{
				const float4 ro = q.ray_o[cur][rbase + t0 + j], rd = q.ray_d[cur][rbase + t0 + j];
				if constexpr (kMode == 2)
				{
					const u64 m = w64 ? (u64)flat_boxes<true>(sc.flat, sc.n_flat, xyz(ro), xyz(rd), 0.001f, JP_INF) : (u64)flat_boxes<false>(sc.flat, sc.n_flat, xyz(ro), xyz(rd), 0.001f, JP_INF);
					s_mask[kMode == 2 ? j : 0] = m;
					key[r] = work_class_flat(__popcll(m));
				}
				else key[r] = work_class_cut(__popc(flat_boxes<false>(sc.cut, sc.n_cut, xyz(ro), xyz(rd), 0.001f, JP_INF)));
			}
		}
		tile_partition<kRPT, JP_SORT_CLASSES>(key, 0x3fu, s_cnt, s_idx);
		// software prefetch: the next pass' ray is requested before this pass' traversal
		float4 ro = make_float4(0, 0, 0, 0), rd = make_float4(0, 0, 1, 0); unsigned int jn = 0;
		// 64-ray chunks, taken by the waves from an LDS counter, most expected work first (the classes sort ascending): the waves
		// reach the barrier at the end of the tile together
		const unsigned int nch = (count + 63u) >> 6, lane = threadIdx.x & 63u;
		unsigned int tk = wave_take(&s_take, 1u), c0 = (nch - 1u - tk) << 6;
		if (tk < nch && c0 + lane < count) { jn = s_idx[c0 + lane]; ro = q.ray_o[cur][rbase + t0 + jn]; rd = q.ray_d[cur][rbase + t0 + jn]; }
		#pragma unroll 1
		while (tk < nch)
		{
			const unsigned int pos = c0 + lane, j = jn;
			const float4 co = ro, cd = rd;
			tk = wave_take(&s_take, 1u); c0 = (nch - 1u - tk) << 6;
			if (tk < nch && c0 + lane < count) { jn = s_idx[c0 + lane]; ro = q.ray_o[cur][rbase + t0 + jn]; rd = q.ray_d[cur][rbase + t0 + jn]; }
			if (pos < count)
			{
				float tmax = JP_INF;                                     // FRay defaults geometry.h:399: min_t 0.001, max_t infinity
				int hit;
				if constexpr (kMode == 2)
				{
					const u64 m = s_mask[kMode == 2 ? j : 0];
					hit = w64 ? flat_prims<false, true, 5>(m, acc.prims, xyz(co), xyz(cd), 0.001f, tmax) : flat_prims<false, false, 5>((unsigned int)m, acc.prims, xyz(co), xyz(cd), 0.001f, tmax);
				}
				else hit = acc.template trace<false>(sc, xyz(co), xyz(cd), 0.001f, tmax);
				q.hit[rbase + t0 + j] = make_float2(tmax, __int_as_float(hit));
				h += hit >= 0 ? 1u : 0u;
			}
		}
		__syncthreads();                                             // s_idx / s_mask are rewritten by the next tile
		if (threadIdx.x == 0) s_take = 0;                            // (the partition's barriers come before the next take)
	}
	for (int off = 32; off > 0; off >>= 1) h += __shfl_down(h, off);
	if ((threadIdx.x & 63) == 0 && h) atomicAdd(&cnt->closest_hit, (unsigned long long)h);
}

// ---------------------------------------------------------------------------------------------------------------------
// k_extend_persist / k_shadow_persist: the traversal kernels of large scenes with LANE REFILL.  There the rays of a wave differ
// wildly in length (most leave the scene after a few nodes, some walk 60+ nodes through a mesh), and a wave that traces one ray
// per lane runs as long as its longest ray: lane utilisation 0.19 / 0.16 on the 280k-triangle scene (profiles/r02b_c3_pmc_sq.txt).
// Here a wave keeps its lanes busy instead: whenever at least kRefill lanes have finished their ray, those lanes deliver their
// result and take the next rays of the workgroup's region from a shared counter in LDS (one wave-aggregated ds_add per refill).
// Every ray is traversed by the same steps in the same order as in traverse / traverse_wide / traverse_ref (Walker<mode>,
// jp_device.h), so hit records, visibility verdicts -- and films -- are unchanged; only WHICH lane traces WHICH ray is.
//   k_extend_persist: one closest-hit ray per lane; the hit record goes to the ray's own queue position.
//   k_shadow_persist: one SHADOW RAY per lane (not one entry): the rays of a region are enumerated plane-major (ray k of entry e =
//     k * E + e) and a ray's verdict is one bit in an LDS bitmap; after a block barrier every thread adds, for the entries it owns,
//     the visible contributions to the path's radiance in light order (integrator.cc:367-370) -- the reference's sum, run-to-run
//     deterministic -- and only those contributions are read from HBM.
// LDS: [stack: stack_cap words per thread, deeper entries in the global spill array (WalkStack)][k_shadow_persist: bitmap of ceil(R * n_planes / 32) words]
// ---------------------------------------------------------------------------------------------------------------------
// One iteration of a refill kernel's wave: the lanes vote on the kind of step it runs (node / leaf) -- the one most active lanes wait for -- and the
// minority sits it out.  (pool: rays are left in the region; only the diagnostic build's statistics use it, tools/turn_stats.py.)
template <int kMode, bool kAnyHit, bool kVote>
__device__ __forceinline__ void persist_turn(Walker<kMode>& w, const SceneView& sc, const WalkStack& stack, bool pool)
{
	const int nh = __popcll(__ballot(!w.done && w.heavy())), nl = __popcll(__ballot(!w.done && !w.heavy()));
	const bool heavyTurn = kVote ? nh > nl : w.heavy();
	if (kVote) { JP_TURN(heavyTurn ? 3 : 0, 1); JP_TURN(heavyTurn ? 4 : 1, heavyTurn ? nh : nl); JP_TURN(heavyTurn ? 5 : 2, nh + nl); if (!pool) { JP_TURN(8, 1); JP_TURN(9, nh + nl); JP_TURN(10, heavyTurn ? nh : nl); } }
	if (!w.done && (w.heavy() == heavyTurn)) w.template step<kAnyHit>(sc, stack);
}
// rank of this lane among the set bits of a wave mask (bits below the lane): v_mbcnt, no 64-bit per-lane mask to keep alive
__device__ __forceinline__ unsigned int lane_rank(unsigned long long m) { return __builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u)); }

template <int kMode, int kRefill, bool kVote>
__global__ void __launch_bounds__(JP_BLOCK, 8) k_extend_persist(SceneView sc, Queues q, int cur_q, int stack_cap, int* spill, DevCounters* cnt)
{
	__shared__ unsigned int s_next;
	const WalkStack stack = { (int*)s_dyn + threadIdx.x, spill + blockIdx.x * JP_BLOCK + threadIdx.x, (kMode == 4 || kMode == 6) ? stack_cap - 1 : stack_cap, gridDim.x * JP_BLOCK };   // Walker<4> / <6>: the last LDS word is the dump slot
	const unsigned int b = blockIdx.x, n = q.blk_q[cur_q][b], rbase = b * q.R;
	if (b == 0 && threadIdx.x == 0) { cnt->closest += cnt->n_queue[cur_q]; cnt->n_queue[cur_q ^ 1] = 0; cnt->n_shadow = 0; }
	if (threadIdx.x == 0) s_next = 0;
	__syncthreads();
	const int lane = threadIdx.x & 63;
	Walker<kMode> w; w.done = true; w.hit = -1; w.tmax = JP_INF;
	unsigned int ridx = 0xffffffffu, h = 0;
	bool pool = n > 0;                                               // wave-uniform: rays may be left in the region
	for (;;)
	{
		const unsigned long long idle = __ballot(w.done);
		const int nidle = __popcll(idle);
		if (nidle == 64 || (pool && nidle >= kRefill))
		{
			if (w.done && ridx != 0xffffffffu)
			{
				int hp = w.hit;
				if constexpr (kMode == 6) { if (!w.certain(sc)) hp = -2; }   // not certified: marked, walked again below the reference's way
				q.hit[rbase + ridx] = make_float2(w.tmax, __int_as_float(hp)); h += hp >= 0 ? 1u : 0u; ridx = 0xffffffffu;
			}
			if (!pool) break;                                        // every lane idle, nothing left
			const int first = __ffsll((long long)idle) - 1;
			unsigned int start = 0;
			if (lane == first) start = atomicAdd(&s_next, (unsigned int)nidle);
			start = __shfl(start, first);
			{ constexpr bool kAnyHit = false; (void)kAnyHit; JP_TURN(6, 1); JP_TURN(7, nidle); }
			pool = start + (unsigned int)nidle < n;
			if (w.done)
			{
				const unsigned int my = start + lane_rank(idle);
				if (my < n)
				{
					const float4 ro = q.ray_o[cur_q][rbase + my], rd = q.ray_d[cur_q][rbase + my];
					w.start(xyz(ro), xyz(rd), 0.001f, JP_INF);       // FRay defaults geometry.h:399
					if constexpr (kMode == 6) w.mark_origin(sc);
					ridx = my;
				}
			}
			if (start >= n && nidle == 64) break;                    // the counter ran past the region while every lane was idle
			continue;
		}
		persist_turn<kMode, false, kVote>(w, sc, stack, pool);
	}
	if constexpr (kMode == 6)
	{   // the marked rays (a few in 10^4), FBVH_Node::Intersect's own walk
		__syncthreads();
		for (unsigned int i = threadIdx.x; i < n; i += JP_BLOCK)
		{
			if (__float_as_int(q.hit[rbase + i].y) != -2) continue;
			const float4 ro = q.ray_o[cur_q][rbase + i], rd = q.ray_d[cur_q][rbase + i];
			float tm = JP_INF;
			const int hp = walk_ray<5, false>(sc, xyz(ro), xyz(rd), 0.001f, tm, stack);
			q.hit[rbase + i] = make_float2(tm, __int_as_float(hp)); h += hp >= 0 ? 1u : 0u;
			if (cnt) atomicAdd(&cnt->cert_fallback, 1ull);
		}
	}
	for (int off = 32; off > 0; off >>= 1) h += __shfl_down(h, off);
	if (lane == 0 && h) atomicAdd(&cnt->closest_hit, (unsigned long long)h);
}

template <int kMode, int kRefill, bool kVote>
__global__ void __launch_bounds__(JP_BLOCK, 8) k_shadow_persist(SceneView sc, Queues q, RenderConst rc, int stack_cap, int* spill, DevCounters* cnt)
{
	__shared__ unsigned int s_next;
	const WalkStack stack = { (int*)s_dyn + threadIdx.x, spill + blockIdx.x * JP_BLOCK + threadIdx.x, (kMode == 4 || kMode == 6) ? stack_cap - 1 : stack_cap, gridDim.x * JP_BLOCK };   // Walker<4> / <6>: the last LDS word is the dump slot
	unsigned int* s_occ = (unsigned int*)s_dyn + stack_cap * JP_BLOCK;   // bit r set: ray r is occluded
	const unsigned int b = blockIdx.x, E = q.blk_sh[b], rbase = b * q.R;
	const unsigned int NP = (unsigned int)rc.n_planes, total = E * NP;
	unsigned int* s_uns = s_occ + ((size_t)q.R * NP + 31) / 32;          // Walker<6>: bit r set: ray r could not be certified
	for (unsigned int i = threadIdx.x; i < (total + 31) / 32; i += JP_BLOCK) { s_occ[i] = 0; if (kMode == 6) s_uns[i] = 0; }
	if (threadIdx.x == 0) s_next = 0;
	__syncthreads();
	const int lane = threadIdx.x & 63;
	Walker<kMode> w; w.done = true; w.hit = -1;
	unsigned int rid = 0xffffffffu;
	bool pool = total > 0;
	for (;;)
	{
		const unsigned long long idle = __ballot(w.done);
		const int nidle = __popcll(idle);
		if (nidle == 64 || (pool && nidle >= kRefill))
		{
			if (w.done && rid != 0xffffffffu)
			{
				if (w.hit >= 0) atomicOr(&s_occ[rid >> 5], 1u << (rid & 31u));
				if constexpr (kMode == 6) { if (w.hit < 0 && w.unsure) atomicOr(&s_uns[rid >> 5], 1u << (rid & 31u)); }
				rid = 0xffffffffu;
			}
			if (!pool) break;
			const int first = __ffsll((long long)idle) - 1;
			unsigned int start = 0;
			if (lane == first) start = atomicAdd(&s_next, (unsigned int)nidle);
			start = __shfl(start, first);
			{ constexpr bool kAnyHit = true; (void)kAnyHit; JP_TURN(6, 1); JP_TURN(7, nidle); }
			pool = start + (unsigned int)nidle < total;
			if (w.done)
			{
				const unsigned int my = start + lane_rank(idle);
				if (my < total)
				{
					const unsigned int k = my / E, e = my - k * E;
					const float4 so = q.sh_o[rbase + e];
					if (k < ((unsigned int)__float_as_int(so.w) >> rc.slot_bits))
					{
						const float4 sd = q.sh_d[(size_t)k * q.cap + rbase + e];
						w.start(xyz(so), xyz(sd), 0.001f, sd.w);     // FScene::Occluded scene.h:36-47
						rid = my;
					}
				}
			}
			if (start >= total && nidle == 64) break;
			continue;
		}
		persist_turn<kMode, true, kVote>(w, sc, stack, pool);
	}
	__syncthreads();
	if constexpr (kMode == 6)
	{   // the rays without a certificate, FBVH_Node::Intersect's own walk
		for (unsigned int i = threadIdx.x; i < (total + 31) / 32; i += JP_BLOCK)
		{
			unsigned int bits = s_uns[i];
			while (bits)
			{
				const unsigned int r = i * 32u + (unsigned int)(__ffs((int)bits) - 1); bits &= bits - 1u;
				const unsigned int k = r / E, e = r - k * E;
				const float4 so = q.sh_o[rbase + e], sd = q.sh_d[(size_t)k * q.cap + rbase + e];
				float tm = sd.w;
				if (walk_ray<5, true>(sc, xyz(so), xyz(sd), 0.001f, tm, stack) >= 0) atomicOr(&s_occ[r >> 5], 1u << (r & 31u));
				if (cnt) atomicAdd(&cnt->cert_fallback, 1ull);
			}
		}
		__syncthreads();
	}
	// ---- the entries' sums, in light order ----
	unsigned int rays = 0, occ = 0;
	for (unsigned int e = threadIdx.x; e < E; e += JP_BLOCK)
	{
		const int packed = __float_as_int(q.sh_o[rbase + e].w);
		const int slot = packed & ((1 << rc.slot_bits) - 1); const unsigned int nr = (unsigned int)packed >> rc.slot_bits;
		unsigned int vm = 0;
		for (unsigned int k = 0; k < nr; k++) { const unsigned int r = k * E + e; if (!((s_occ[r >> 5] >> (r & 31u)) & 1u)) vm |= 1u << k; }
		rays += nr; occ += nr - (unsigned int)__popc(vm);
		if (vm)
		{
			const float4 L = q.lacc[slot];
			V3 a = mk(L.x, L.y, L.z);
			for (unsigned int k = 0; k < nr; k++)
				if ((vm >> k) & 1u) { const float4 c4 = q.sh_c[(size_t)k * q.cap + rbase + e]; a = a + xyz(c4); }
			q.lacc[slot] = make_float4(a.x, a.y, a.z, 0.f);
		}
	}
	for (int off = 32; off > 0; off >>= 1) { rays += __shfl_down(rays, off); occ += __shfl_down(occ, off); }
	if (lane == 0) { if (rays) atomicAdd(&cnt->shadow, (unsigned long long)rays); if (occ) atomicAdd(&cnt->shadow_occ, (unsigned long long)occ); }
}

// k_shadow_sort: k_shadow with the ENTRIES of each tile partitioned by the expected work of their rays (summed over the entry's
// rays).  One lane still owns an entry and adds its visible contributions in light order (integrator.cc:367-370).
template <int kMode>
__global__ void __launch_bounds__(JP_BLOCK, 4) k_shadow_sort(SceneView sc, Queues q, RenderConst rc, int depth, DevCounters* cnt)
{
	constexpr int kRPT = JP_SORT_TILE / JP_BLOCK;
	__shared__ unsigned short s_idx[JP_SORT_TILE];
	__shared__ unsigned int s_cnt[JP_SORT_CLASSES * kRPT * (JP_BLOCK / 64)];
	__shared__ u64 s_mask[kMode == 2 ? 2 * JP_SORT_TILE : 1];       // mode 2: box-phase masks of the entry's first two rays (further rays redo the phase)
	__shared__ unsigned int s_take;
	if (threadIdx.x == 0) s_take = 0;
	SceneAccess<kMode> acc(sc, depth);
	const unsigned int b = blockIdx.x, E = q.blk_sh[b], rbase = b * q.R;
	const bool w64 = sc.n_prims > 32;
	unsigned int rays = 0, occ = 0;
	for (unsigned int e0 = 0; e0 < E; e0 += JP_SORT_TILE)
	{
		const unsigned int count = E - e0 < (unsigned int)JP_SORT_TILE ? E - e0 : (unsigned int)JP_SORT_TILE;
		unsigned int key[kRPT];
		#pragma unroll
		for (int r = 0; r < kRPT; r++)
		{
			const unsigned int j = r * JP_BLOCK + threadIdx.x;
			key[r] = JP_SORT_CLASSES;
			if (j < count)
			{
				const float4 so = q.sh_o[rbase + e0 + j];
				const int nr = (int)((unsigned int)__float_as_int(so.w) >> rc.slot_bits);
				int pc = 0;
				for (int k = 0; k < nr; k++)
				{
					const float4 sd = q.sh_d[(size_t)k * q.cap + rbase + e0 + j];
					if constexpr (kMode == 2)
					{
						if (k < 2)
						{
							const u64 m = w64 ? (u64)flat_boxes<true>(sc.flat, sc.n_flat, xyz(so), xyz(sd), 0.001f, sd.w) : (u64)flat_boxes<false>(sc.flat, sc.n_flat, xyz(so), xyz(sd), 0.001f, sd.w);
							s_mask[kMode == 2 ? k * JP_SORT_TILE + j : 0] = m;
							pc += __popcll(m);
						}
					}
					else pc += __popc(flat_boxes<false>(sc.cut, sc.n_cut, xyz(so), xyz(sd), 0.001f, sd.w));
				}
				key[r] = kMode == 2 ? work_class_flat(pc) : work_class_cut(pc);
			}
		}
		tile_partition<kRPT, JP_SORT_CLASSES>(key, 0x3fu, s_cnt, s_idx);
		float4 so_n = make_float4(0, 0, 0, 0), sd_n = make_float4(0, 0, 1, 0); unsigned int jn = 0;
		const unsigned int nch = (count + 63u) >> 6, lane = threadIdx.x & 63u;      // chunks taken most work first, as in k_extend_sort
		unsigned int tk = wave_take(&s_take, 1u), c0 = (nch - 1u - tk) << 6;
		if (tk < nch && c0 + lane < count) { jn = s_idx[c0 + lane]; so_n = q.sh_o[rbase + e0 + jn]; sd_n = q.sh_d[rbase + e0 + jn]; }
		#pragma unroll 1
		while (tk < nch)
		{
			const unsigned int pos = c0 + lane, j = jn, e = rbase + e0 + j;
			const float4 so = so_n; float4 sd = sd_n;
			tk = wave_take(&s_take, 1u); c0 = (nch - 1u - tk) << 6;
			if (tk < nch && c0 + lane < count) { jn = s_idx[c0 + lane]; so_n = q.sh_o[rbase + e0 + jn]; sd_n = q.sh_d[rbase + e0 + jn]; }
			if (pos >= count) continue;
			const int packed = __float_as_int(so.w);
			const int slot = packed & ((1 << rc.slot_bits) - 1), nr = (int)((unsigned int)packed >> rc.slot_bits);
			if (nr == 0) continue;
			bool any = false;
			const float4 L = q.lacc[slot];                               // issued up front: its latency hides behind the traversal
			V3 a = mk(L.x, L.y, L.z);
			for (int k = 0; k < nr; k++)
			{
				const float4 c4 = q.sh_c[(size_t)k * q.cap + e];          // needed only after the traversal
				float tmax = sd.w;
				const V3 dir = xyz(sd);
				if (k + 1 < nr) sd = q.sh_d[(size_t)(k + 1) * q.cap + e];
				int hit = -1;
				bool traced = false;
				if constexpr (kMode == 2)
				{
					if (k < 2)
					{
						const u64 m = s_mask[k * JP_SORT_TILE + j];
						hit = w64 ? flat_prims<true, true, 5>(m, acc.prims, xyz(so), dir, 0.001f, tmax) : flat_prims<true, false, 5>((unsigned int)m, acc.prims, xyz(so), dir, 0.001f, tmax);
						traced = true;
					}
				}
				if (!traced) hit = acc.template trace<true>(sc, xyz(so), dir, 0.001f, tmax);
				rays++;
				if (hit >= 0) occ++;
				else { a = a + xyz(c4); any = true; }
			}
			if (any) q.lacc[slot] = make_float4(a.x, a.y, a.z, 0.f);
		}
		__syncthreads();                                             // s_idx / s_mask are rewritten by the next tile
		if (threadIdx.x == 0) s_take = 0;
	}
	for (int off = 32; off > 0; off >>= 1) { rays += __shfl_down(rays, off); occ += __shfl_down(occ, off); }
	if ((threadIdx.x & 63) == 0) { if (rays) atomicAdd(&cnt->shadow, (unsigned long long)rays); if (occ) atomicAdd(&cnt->shadow_occ, (unsigned long long)occ); }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_other: the reference's other two integrators behind the same Render() seam, as a megakernel --
// one thread owns one camera sample and walks it to the end; not the hot path, no queues.
//   FWhittedIntegrator::Li integrator.cc:115-220 branches: a mirror matches SpecularReflect AND SpecularReflectAndTransmit
//   (MatchTypes, bsdf.h:282), glass matches the latter, and the sampler draws of the second branch start where the first
//   branch's subtree stopped -- so a sample is walked depth first by one thread with an explicit frame stack, which yields
//   the reference's draw order and its nested products f * Li(child) * |cos| / pdf by construction.
//   FDebugIntegrator::Li integrator.h:44-58: the hit normal as a colour.
// The radiance goes to lacc[slot]; k_resolve sums the samples of a pixel as for the path integrator.
// ---------------------------------------------------------------------------------------------------------------------
#define JP_WHITTED_MAX_DEPTH 16
struct WFrame { V3 L, p, N, wo, f; float absd, pdf, up; int mat, k; };

template <int kMode>
__global__ void __launch_bounds__(JP_BLOCK) k_other(SceneView sc, Queues q, RenderConst rc, int kind, int depth_lds, DevCounters* cnt)
{
	SceneAccess<kMode> acc(sc, depth_lds);
	const unsigned int total = (unsigned int)rc.npix * rc.sbatch;
	unsigned int n_closest = 0, n_hit = 0, n_shadow = 0, n_occ = 0;
	for (unsigned int slot = blockIdx.x * JP_BLOCK + threadIdx.x; slot < total; slot += gridDim.x * JP_BLOCK)
	{
		const int pix = slot % rc.npix, s = rc.s0 + slot / rc.npix;
		int x, y; pixel_of(rc, pix, x, y);
		const uint32_t key = jp_rng_key(rc.seed, (uint32_t)x, (uint32_t)y, (uint32_t)s);
		unsigned int dim = 2;
		V3 o, d;
		{
			const float fx = (float)x + rngf(rc, key, 0), fy = (float)y + rngf(rc, key, 1);
			const V3 front = mk(sc.cam.front[0], sc.cam.front[1], sc.cam.front[2]), right = mk(sc.cam.right[0], sc.cam.right[1], sc.cam.right[2]), up = mk(sc.cam.up[0], sc.cam.up[1], sc.cam.up[2]);
			o = mk(sc.cam.pos[0], sc.cam.pos[1], sc.cam.pos[2]);
			d = normalize(front + right * (fx / sc.cam.res_x - 0.5f) + up * (0.5f - fy / sc.cam.res_y));
		}
		V3 result = mk(0, 0, 0);
		if (kind == JP_INTEGRATOR_DEBUG_NORMAL)
		{
			float tmax = JP_INF; const int hit = acc.template trace<false>(sc, o, d, 0.001f, tmax);
			n_closest++;
			if (hit >= 0)
			{
				n_hit++;
				const float4 g3 = sc.prims[4 * hit + 3]; const int type = __float_as_int(g3.w);
				const V3 p = o + tmax * d;
				if (type == JP_SHAPE_TRIANGLE) result = xyz(g3);
				else if (type == JP_SHAPE_RECTANGLE) result = dot(xyz(g3), d) <= 0 ? xyz(g3) : -xyz(g3);
				else if (type == JP_SHAPE_DISK) result = xyz(sc.prims[4 * hit + 1]);
				else result = normalize(p - xyz(sc.prims[4 * hit]));
			}
			q.lacc[slot] = make_float4(result.x, result.y, result.z, 0.f);
			continue;
		}
		WFrame st[JP_WHITTED_MAX_DEPTH];
		int sp = 0;
		V3 ret = mk(0, 0, 0);
		bool entering = true;
		for (;;)
		{
			if (entering)
			{   // ---- Li(ray, depth = sp): intersection, emission, direct light (integrator.cc:119-158) ----
				int hit, mat = -1; float tmax; V3 p = o, N = mk(0, 0, 1);
				for (;;)
				{
					tmax = JP_INF; hit = acc.template trace<false>(sc, o, d, 0.001f, tmax);
					n_closest++;
					if (hit < 0) break;
					n_hit++;
					const float4 g3 = sc.prims[4 * hit + 3]; const int type = __float_as_int(g3.w);
					p = o + tmax * d;
					if (type == JP_SHAPE_TRIANGLE) N = xyz(g3);
					else if (type == JP_SHAPE_RECTANGLE) N = dot(xyz(g3), d) <= 0 ? xyz(g3) : -xyz(g3);
					else if (type == JP_SHAPE_DISK) N = xyz(sc.prims[4 * hit + 1]);
					else N = normalize(p - xyz(sc.prims[4 * hit]));
					mat = sc.meta[hit].y;
					if (mat >= 0) break;
					o = p;                                                            // nullptr material: same direction, same depth (integrator.cc:137-139)
				}
				if (hit < 0) { ret = sc.n_env > 0 ? mk(sc.env_sum.x, sc.env_sum.y, sc.env_sum.z) : mk(0, 0, 0); entering = false; continue; }   // integrator.cc:123-128
				const int mtype = sc.mat_type[mat];
				float up = 0.f; if (mtype == JP_MAT_PLASTIC) up = rngf(rc, key, dim++);
				Closure c; make_closure(sc.mats, mtype, mat, up, c);
				const Frame fr = frame_from_z(N);
				const V3 wo_w = -d, wo = to_local(fr, wo_w);
				closure_set_wo(c, wo);
				V3 L = mk(0, 0, 0);
				{
					const int li = sc.meta[hit].z;
					V3 Le = mk(0, 0, 0);
					if (li >= 0 && dot(N, wo_w) > 0.f) Le = xyz(sc.lights[2 * li]);
					L = L + Le;                                                       // integrator.cc:142
				}
				for (int li = 0; li < sc.n_lights; li++)                              // integrator.cc:145-158
				{
					const float ux = rngf(rc, key, dim), uy = rngf(rc, key, dim + 1); dim += 2;
					const float4 lrad = sc.lights[2 * li];
					LightSample ls = sample_li(sc, sc.prims, sc.lights, li, p, N, ux, uy);
					if (isblack(ls.Li) || ls.pdf == 0.f) continue;
					const V3 f = eval_local(c, wo, to_local(fr, ls.wi));
					if (isblack(f)) continue;
					const V3 sdir = __float_as_int(lrad.w) == JP_LIGHT_AREA ? ls.wi : normalize(ls.pos - p);
					float stmax = len(p - ls.pos) - 0.001f;
					n_shadow++;
					if (acc.template trace<true>(sc, p, sdir, 0.001f, stmax) >= 0) { n_occ++; continue; }
					L = L + cmul(f, ls.Li) * absdot(ls.wi, N) / ls.pdf;
				}
				st[sp].L = L; st[sp].p = p; st[sp].N = N; st[sp].wo = wo_w; st[sp].mat = mat; st[sp].up = up; st[sp].k = 0;
				entering = false;
				ret = mk(0, 0, 0);
				// fall into the branch loop of this frame with nothing to add yet
				st[sp].f = mk(0, 0, 0); st[sp].absd = 0.f; st[sp].pdf = 1.f;
				goto branch;
			}
			// ---- a value `ret` comes back: finished sample, or the child of the frame below ----
			if (sp == 0) { result = ret; break; }
			sp--;
			st[sp].L = st[sp].L + cmul(st[sp].f, ret) * st[sp].absd / st[sp].pdf;       // integrator.cc:185, 203, 220
		branch:
			{
				WFrame& F = st[sp];
				bool descended = false;
				if (sp + 1 < rc.max_depth)                                            // integrator.cc:161
				{
					const int mtype = sc.mat_type[F.mat];
					Closure c; make_closure(sc.mats, mtype, F.mat, F.up, c);
					const int flags = c.kind == CL_LAMBERT ? (1 | 8) : c.kind == CL_MIRROR ? (1 | 4) : c.kind == CL_FRESNEL_SPECULAR ? (4 | 1 | 2) : (1 | 16);
					while (F.k < 3)
					{
						const int want = F.k == 0 ? (4 | 1) : (F.k == 1 ? (4 | 2) : (4 | 1 | 2));   // SpecularReflect / Transmit / ReflectAndTransmit
						F.k++;
						if ((flags & want) != flags) continue;                         // MatchTypes bsdf.h:282
						const float ux = rngf(rc, key, dim), uy = rngf(rc, key, dim + 1); dim += 2;
						const Frame fr = frame_from_z(F.N);
						const V3 wol = to_local(fr, F.wo);
						closure_set_wo(c, wol);
						BsdfSample bs = sample_local(c, wol, ux, uy);
						bs.wi = to_world(fr, bs.wi);
						if (isblack(bs.f) || bs.pdf == 0.f) continue;
						F.f = bs.f; F.absd = absdot(bs.wi, F.N); F.pdf = bs.pdf;
						o = F.p; d = bs.wi;
						sp++; entering = true; descended = true;
						break;
					}
				}
				if (!descended) ret = F.L;                                            // integrator.cc:169: this frame is done
			}
		}
		q.lacc[slot] = make_float4(result.x, result.y, result.z, 0.f);
	}
	for (int off = 32; off > 0; off >>= 1) { n_closest += __shfl_down(n_closest, off); n_hit += __shfl_down(n_hit, off); n_shadow += __shfl_down(n_shadow, off); n_occ += __shfl_down(n_occ, off); }
	if ((threadIdx.x & 63) == 0)
	{
		if (n_closest) atomicAdd(&cnt->closest, (unsigned long long)n_closest);
		if (n_hit) atomicAdd(&cnt->closest_hit, (unsigned long long)n_hit);
		if (n_shadow) atomicAdd(&cnt->shadow, (unsigned long long)n_shadow);
		if (n_occ) atomicAdd(&cnt->shadow_occ, (unsigned long long)n_occ);
	}
}

// ---------------------------------------------------------------------------------------------------------------------
// k_resolve: the per-pixel sample loop's sum (integrator.cc:89,102-108) in sample-index order; the running sum of a
// pixel lives in pix_acc across batches; the last batch writes Clamp01 onto the (zero) film.
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(JP_BLOCK) k_resolve(Queues q, RenderConst rc, float4* pix_acc, float* film, int first, int last)
{
	const float ratio = 1.0f / (float)rc.spp;
	for (int pix = blockIdx.x * JP_BLOCK + threadIdx.x; pix < rc.npix; pix += gridDim.x * JP_BLOCK)
	{
		V3 L = mk(0, 0, 0);
		if (!first) { const float4 a = pix_acc[pix]; L = mk(a.x, a.y, a.z); }
		for (int s = 0; s < rc.sbatch; s++)
		{
			const float4 l = q.lacc[(size_t)s * rc.npix + pix];
			L = L + mk(l.x, l.y, l.z) * ratio;
		}
		if (!last) pix_acc[pix] = make_float4(L.x, L.y, L.z, 0.f);
		else
		{
			int x, y; pixel_of(rc, pix, x, y);
			float* o = film + 3 * ((size_t)y * rc.width + x);
			o[0] = 0.f + clampf(L.x, 0.f, 1.f); o[1] = 0.f + clampf(L.y, 0.f, 1.f); o[2] = 0.f + clampf(L.z, 0.f, 1.f);   // film.h:22-23, 64-68
		}
	}
}

#include "jp_path.h"

// ---------------------------------------------------------------------------------------------------------------------
// k_tonemap8: gamma_encoding (film.h:24) of the resolved film on the device -> 3 bytes per pixel for the BMP / PPM writers of
// FFilm::SaveAsImage (film.cc:45-145).  thr[k-1] is the smallest fp32 x in [0, 1] whose host-side gamma_encoding(x) is >= k
// (host_gamma_thresholds); the byte is the number of thresholds <= x, found by an 8-step binary search in LDS -- identical
// to the host's powf-based value for every fp32 input by construction, without reproducing powf on the device.
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(JP_BLOCK) k_tonemap8(const float* __restrict__ film, unsigned char* __restrict__ rgb8, const float* __restrict__ thr, size_t n)
{
	__shared__ float s_thr[256];
	s_thr[threadIdx.x] = threadIdx.x < 255 ? thr[threadIdx.x] : JP_INF;
	__syncthreads();
	for (size_t i = (size_t)blockIdx.x * JP_BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * JP_BLOCK)
	{
		const float x = clampf(film[i], 0.f, 1.f);                  // Clamp01 (the film is clamped already; NaN cannot occur)
		int lo = 0;                                                  // number of thresholds known to be <= x
		#pragma unroll
		for (int step = 128; step > 0; step >>= 1) if (lo + step <= 255 && s_thr[lo + step - 1] <= x) lo += step;
		rgb8[i] = (unsigned char)lo;
	}
}

// ---------------------------------------------------------------------------------------------------------------------
// k_bsdf: FBSDF::Evalf / Pdf / Sample (bsdf.h:284-302) of a by-value BSDF (jp_xbsdf.h) for n shading events -- behind jp_bsdf
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(JP_BLOCK) k_bsdf(JpBsdfDesc d, int n, const float* nrm, const float* wo, const float* wi, const float* u,
                                                   float* feval, float* pdfeval, float* sf, float* swi, float* spdf, int* sflags)
{
	for (int i = blockIdx.x * JP_BLOCK + threadIdx.x; i < n; i += gridDim.x * JP_BLOCK)
	{
		const Frame fr = frame_from_z(mk(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]));
		const V3 wol = to_local(fr, mk(wo[3 * i], wo[3 * i + 1], wo[3 * i + 2])), wil = to_local(fr, mk(wi[3 * i], wi[3 * i + 1], wi[3 * i + 2]));
		const V3 f = xb::x_eval(d, fr, wol, wil);
		const float pe = xb::x_pdf(d, fr, wol, wil);
		BsdfSample s = xb::x_sample(d, fr, wol, u[2 * i], u[2 * i + 1]);
		s.wi = to_world(fr, s.wi);
		feval[3 * i] = f.x; feval[3 * i + 1] = f.y; feval[3 * i + 2] = f.z; pdfeval[i] = pe;
		sf[3 * i] = s.f.x; sf[3 * i + 1] = s.f.y; sf[3 * i + 2] = s.f.z;
		swi[3 * i] = s.wi.x; swi[3 * i + 1] = s.wi.y; swi[3 * i + 2] = s.wi.z;
		spdf[i] = s.pdf; sflags[i] = s.flags;
	}
}

// ---------------------------------------------------------------------------------------------------------------------
// k_trace: test hook, arbitrary rays through the same traversal
// ---------------------------------------------------------------------------------------------------------------------
template <int kMode>
__global__ void __launch_bounds__(JP_BLOCK) k_trace(SceneView sc, int depth, int n, const float* o, const float* d, const float* tmin, const float* tmax_in,
                                                    int* hit, float* t, int* prim, float* nrm)
{
	SceneAccess<kMode> acc(sc, depth);
	for (int i = blockIdx.x * JP_BLOCK + threadIdx.x; i < n; i += gridDim.x * JP_BLOCK)
	{
		const V3 ro = mk(o[3 * i], o[3 * i + 1], o[3 * i + 2]), rd = mk(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
		float tmax = tmax_in[i];
		const int h = acc.template trace<false>(sc, ro, rd, tmin[i], tmax);
		hit[i] = h >= 0; t[i] = tmax; prim[i] = h >= 0 ? sc.meta[h].x : -1;
		V3 N = mk(0, 0, 0);
		if (h >= 0)
		{
			const float4 g3 = sc.prims[4 * h + 3]; const int type = __float_as_int(g3.w);
			const V3 p = ro + tmax * rd;
			if (type == JP_SHAPE_TRIANGLE) N = xyz(g3);
			else if (type == JP_SHAPE_RECTANGLE) N = dot(xyz(g3), rd) <= 0 ? xyz(g3) : -xyz(g3);
			else if (type == JP_SHAPE_DISK) N = xyz(sc.prims[4 * h + 1]);
			else N = normalize(p - xyz(sc.prims[4 * h]));
		}
		nrm[3 * i] = N.x; nrm[3 * i + 1] = N.y; nrm[3 * i + 2] = N.z;
	}
}

// =====================================================================================================================
// host side: context, scene upload, render loop, C ABI
// =====================================================================================================================
static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIP_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return fail(JP_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)); } while (0)

struct JpContext
{
	int device = 0;
	hipStream_t stream = nullptr;
	int n_cus = 256;
	// scene
	bool have_scene = false;
	SceneView sv; int stack_depth = 1; bool scene_in_lds = false, shade_prims_in_lds = false; size_t lds_bytes = 0, lds_bytes_shadow = 0;
	void *d_flat = nullptr, *d_wide = nullptr, *d_cut = nullptr, *d_q4 = nullptr; bool ray_sort = false; int trav_mode = 0;
	// k_shadow of bounce b and k_extend of bounce b + 1 both depend on k_shade of bounce b only: with `dual` the shadow launches go to a second
	// stream of the lane (own spill area) and run beside the next extend launch; the next k_shade waits for both
	bool dual = false; hipStream_t stream2 = nullptr; hipEvent_t ev_shade = nullptr, ev_shadow = nullptr; int* d_spill2 = nullptr; size_t spill2_words = 0;
	size_t trav_lds_pad = 0;                                                           // experiment: extra dynamic LDS of the refill kernels = fewer of their workgroups per CU (room for another lane's k_shade)
	void* d_refbox = nullptr; bool cert = false; int cert_eye_leaves = 0;                                       // reference semantics, certified walk (Walker<6>): leaf boxes per primitive
	bool use_q4 = false, q4_shadow = false;                                            // closest-hit (and, as an experiment, shadow) rays walk the 4-wide quantised tree (Walker<4>)
	bool vote = false; int persist = 0;                                             // lane refill in the closest-hit traversal of large scenes (k_extend_persist): refill threshold, 0 = off
	void *d_nodes = nullptr, *d_prims = nullptr, *d_meta = nullptr, *d_mats = nullptr, *d_mat_type = nullptr, *d_lights = nullptr, *d_shade_tab = nullptr;
	int n_planes = 1; bool has_null_material = false; int sincosf_mode = 0, libm_mode = 0;
	bool build_on_device = false; float build_ms = 0.f; int bvh_height = 0, bvh_nodes = 0;
	bool tables_in_lds = false, stage_nee = false; size_t shade_lds_bytes = 0;
	int class_mask = 0x3f; bool shade_sort = false;                                     // k_shade partitions its tiles by material class (scenes with more than one material kind)
	// queues
	Queues q = {}; unsigned int cap = 0; int planes_alloc = 0; unsigned int blk_alloc = 0; int blocks_per_cu = 16;
	std::vector<void*> qbufs;
	float4* d_pix_acc = nullptr; size_t pix_acc_n = 0;
	// lane refill kernels: traversal-stack words per thread kept in LDS, the rest spills to global memory (WalkStack).  Measured on the
	// 280k-triangle scene (tree height 24): 8 / 12 / 16 words 1922 / 1926 / 1922 Msamples/s, 20 words or the whole stack 1634 / 1692.
	int stack_lds_words = 12;
	int* d_spill = nullptr; size_t spill_words = 0;
	float* d_film = nullptr; size_t film_n = 0;
	float *d_bsdf_in = nullptr, *d_bsdf_out = nullptr; int* d_bsdf_fl = nullptr; size_t bsdf_cap = 0;   // jp_bsdf scratch
	float* d_gamma = nullptr; unsigned char* d_rgb8 = nullptr; size_t rgb8_n = 0; unsigned char* h_rgb8 = nullptr; size_t h_rgb8_n = 0;   // jp_render_rgb8
	float* h_film = nullptr; size_t h_film_n = 0;                // pinned staging buffer of jp_render (a pageable copy of the film costs ~2 ms)
	DevCounters* d_cnt = nullptr;
	// timing
	hipEvent_t ev0 = nullptr, ev1 = nullptr;
	bool profiling = false;
	std::vector<hipEvent_t> evpool; size_t evused = 0;
	struct Stamp { int cls; size_t a, b; };
	std::vector<Stamp> stamps;
	JpCounters counters;
	// Extra "lanes": the shard's bands are dealt round-robin to L lanes (this context + L - 1 lane contexts) and rendered
	// concurrently on L streams with L queue sets, so the tail and the launch gap of one lane's kernel are filled by another
	// lane's and bandwidth-bound kernels overlap instruction-bound ones ("Stream lanes").  A lane shares
	// the scene tables (not owned) and writes its bands into its own film; the films are merged at the end.
	std::vector<JpContext*> lanes; bool is_lane = false; unsigned long long own_samples = 0; bool bpc_from_env = false;
	hipEvent_t ev_added = nullptr; bool added_valid = false; int last_lanes = 1;      // lanes used by the last render (1: this context alone)
	// fused schedule (k_path, jp_path.h): region queues of the resident workgroups, the batch's radiance array, job counters
	Queues fq = {}; std::vector<void*> fbufs; unsigned int fcap = 0; int fplanes = 0; size_t flacc_n = 0;
	unsigned int* d_jobs = nullptr; size_t jobs_n = 0;
	int last_fused = 0, last_region = 0, last_wgs = 0;
};

static void free_scene(JpContext* c)
{
	void** ps[] = { &c->d_flat, &c->d_cut, &c->d_wide, &c->d_q4, &c->d_refbox, &c->d_nodes, &c->d_prims, &c->d_meta, &c->d_mats, &c->d_mat_type, &c->d_lights, &c->d_shade_tab };
	for (void** p : ps) { if (*p) hipFree(*p); *p = nullptr; }
	c->have_scene = false;
}
static void free_queues(JpContext* c)
{
	for (void* p : c->qbufs) hipFree(p);
	c->qbufs.clear(); c->cap = 0; c->planes_alloc = 0;
}
static void free_fused(JpContext* c)
{
	for (void* p : c->fbufs) hipFree(p);
	c->fbufs.clear(); c->fcap = 0; c->fplanes = 0; c->flacc_n = 0;
	if (c->d_jobs) hipFree(c->d_jobs); c->d_jobs = nullptr; c->jobs_n = 0;
}

// Which build of glibc's sinf / cosf / sincosf does this host run (jp_shading.h, sincosf_libm)?  The reference computes its
// bounce directions with them, so the device reproduces whichever the host's IFUNC resolver picked: 1 = the FMA build,
// 2 = the build without contraction, 0 = neither reproduces the host on the probe set (another libm): the device then
// keeps its own correctly rounded evaluation.
static int probe_host_sincosf()
{
	static int cached = -1;
	if (cached >= 0) return cached;
	bool okF = true, okN = true;
	uint32_t st = 0x12345u;
	for (int i = 0; i < 200000 && (okF || okN); i++)
	{
		st = st * 1664525u + 1013904223u;
		float y;
		if (i < 150000) y = (float)(st >> 8) * (1.0f / 16777216.0f) * 6.2831855f;      // the call sites' range [0, 2 pi)
		else if (i < 180000) y = (float)(st >> 8) * (1.0f / 16777216.0f) * 0.01f;       // small arguments, incl. the < 2^-12 branch
		else y = ((float)(st >> 8) * (1.0f / 16777216.0f) - 0.5f) * 200.f;               // both signs, up to |y| = 100
		float hs, hc; ::sincosf(y, &hs, &hc);
		const float h1 = ::sinf(y), h2 = ::cosf(y);
		float as, ac, bs, bc;
		jp::sincosf_libm<true>(y, &as, &ac); jp::sincosf_libm<false>(y, &bs, &bc);
		uint32_t uhs, uhc, u1, u2, uas, uac, ubs, ubc;
		std::memcpy(&uhs, &hs, 4); std::memcpy(&uhc, &hc, 4); std::memcpy(&u1, &h1, 4); std::memcpy(&u2, &h2, 4);
		std::memcpy(&uas, &as, 4); std::memcpy(&uac, &ac, 4); std::memcpy(&ubs, &bs, 4); std::memcpy(&ubc, &bc, 4);
		if (uhs != u1 || uhc != u2) { okF = okN = false; }                             // sinf / cosf / sincosf must agree with each other
		if (uas != uhs || uac != uhc) okF = false;
		if (ubs != uhs || ubc != uhc) okN = false;
	}
	cached = okF ? 1 : (okN ? 2 : 0);
	if (const char* e = getenv("JETPBRT_SINCOSF")) { int v = atoi(e); if (v >= 0 && v <= 2) cached = v; }
	return cached;
}

// Do jp_libm.h's logf / expf / powf / acosf / atanf / tanf reproduce the host's libm (jp_xbsdf.h: g_libm_mode)?  Bit 0: all six do on
// every probe argument; bit 1: with the FMA build of the first three (glibc's IFUNC picks it on CPUs with FMA + AVX2; the two builds
// differ on about one argument in 10^8, so the CPU feature decides and the probe confirms).  0: another libm -- the device keeps its
// own library for these functions (k_bsdf then matches the reference within the tolerance of tests/test_gpu_parity.py, not bit for bit).
static int probe_host_libm()
{
	static int cached = -1;
	if (cached >= 0) return cached;
	bool fma_cpu = false;
#if defined(__x86_64__)
	fma_cpu = __builtin_cpu_supports("fma") && __builtin_cpu_supports("avx2");
#endif
	auto same = [](float a, float b) { uint32_t x, y; std::memcpy(&x, &a, 4); std::memcpy(&y, &b, 4); return x == y || (a != a && b != b); };
	auto run = [&](bool fmab) {
		uint32_t st = 0x2545f491u;
		auto rnd = [&]() { st = st * 1664525u + 1013904223u; return st; };
		auto u01 = [&]() { return (float)(rnd() >> 8) * (1.0f / 16777216.0f); };
		for (int i = 0; i < 200000; i++)
		{
			float x, y;
			switch (i & 3)
			{
			case 0: { const uint32_t a = rnd(), b = rnd(); std::memcpy(&x, &a, 4); std::memcpy(&y, &b, 4); break; }   // raw bit patterns: every exponent, specials
			case 1: x = u01(); y = u01() * 8.f; break;                                                                  // the call sites' ranges
			case 2: x = (u01() - 0.5f) * 250.f; y = (u01() - 0.5f) * 64.f; break;
			default: x = u01() * 1e-3f; y = 1.f / (u01() * 100.f + 1.f); break;
			}
			const float e = fmab ? jp::lm::expf_libm<true>(x) : jp::lm::expf_libm<false>(x), l = fmab ? jp::lm::logf_libm<true>(x) : jp::lm::logf_libm<false>(x);
			const float pw = fmab ? jp::lm::powf_libm<true>(x, y) : jp::lm::powf_libm<false>(x, y);
			if (!same(e, ::expf(x)) || !same(l, ::logf(x)) || !same(pw, ::powf(x, y))) return false;
			const float a = (i & 3) == 0 ? x : x * 2.f - 1.f;
			if (!same(jp::lm::acosf_libm(a), ::acosf(a)) || !same(jp::lm::atanf_libm(x), ::atanf(x))) return false;
			bool ok; const float t = jp::lm::tanf_libm(x * 8.f, &ok);
			if (ok && !same(t, ::tanf(x * 8.f))) return false;
		}
		return true;
	};
	int mode = 0;
	if (run(fma_cpu)) mode = 1 | (fma_cpu ? 2 : 0);
	else if (run(!fma_cpu)) mode = 1 | (fma_cpu ? 0 : 2);
	if (const char* e = getenv("JETPBRT_LIBM")) { int v = atoi(e); if (v >= 0 && v <= 3) mode = v; }
	cached = mode;
	return cached;
}

// gamma_encoding of film.h:24 exactly as the host computes it (std::pow on floats = powf, product in double, truncation)
static inline unsigned char host_gamma_encoding(float x)
{
	const float c = x < 0.f ? 0.f : (x > 1.f ? 1.f : x);
	return (unsigned char)(std::pow(c, (float)(1 / 2.2)) * 255.0);
}
// thr[k-1] = smallest fp32 x in [0, 1] with host_gamma_encoding(x) >= k, k = 1..255: the floats of [0, 1] are ordered like
// their bit patterns and the encoding is non-decreasing, so each threshold is a binary search over 0 .. 0x3f800000
static const float* host_gamma_thresholds()
{
	static float thr[255]; static std::once_flag once;
	std::call_once(once, []() {
		for (int k = 1; k <= 255; k++)
		{
			uint32_t lo = 0, hi = 0x3f800000u;                       // enc(lo) < k (enc(0) = 0) ... enc(hi) >= k (enc(1) = 255)
			while (hi - lo > 1) { const uint32_t mid = lo + (hi - lo) / 2; float f; std::memcpy(&f, &mid, 4); if (host_gamma_encoding(f) >= k) hi = mid; else lo = mid; }
			std::memcpy(&thr[k - 1], &hi, 4);
		}
	});
	return thr;
}
// The binary search above assumes that the host's powf-based encoding never steps DOWN on [0, 1] (powf is accurate to under an ulp, not
// guaranteed monotone).  This sweeps EVERY float bit pattern of [0, 1] -- 1,065,353,217 values, n_threads host threads -- and counts
// the values whose byte differs from (number of thresholds <= x): 0 means the device tone map is byte-identical to gamma_encoding for
// every input (tests/test_host_and_abi.py).
static unsigned long long host_gamma_sweep(int n_threads)
{
	const float* thr = host_gamma_thresholds();
	n_threads = std::max(1, std::min(64, n_threads));
	std::vector<unsigned long long> bad((size_t)n_threads, 0ull);
	std::vector<std::thread> pool;
	const uint64_t total = 0x3f800000ull + 1;
	for (int t = 0; t < n_threads; t++)
		pool.emplace_back([&, t]() {
			const uint64_t a = total * t / n_threads, b = total * (t + 1) / n_threads;
			int k = 0;                                                    // thresholds <= x: x ascends, so k only grows
			{ const uint32_t u = (uint32_t)a; float f; std::memcpy(&f, &u, 4); while (k < 255 && thr[k] <= f) k++; }
			unsigned long long nb = 0;
			for (uint64_t i = a; i < b; i++)
			{
				const uint32_t u = (uint32_t)i; float f; std::memcpy(&f, &u, 4);
				while (k < 255 && thr[k] <= f) k++;
				if (host_gamma_encoding(f) != (unsigned char)k) nb++;
			}
			bad[(size_t)t] = nb;
		});
	for (auto& th : pool) th.join();
	unsigned long long s2 = 0; for (unsigned long long v : bad) s2 += v;
	return s2;
}

extern "C" {

const char* jp_last_error(void) { return g_err.c_str(); }
int jp_gamma_thresholds(float* out255) { if (!out255) return fail(JP_ERR_INVALID_ARGUMENT, "jp_gamma_thresholds: null argument"); std::memcpy(out255, host_gamma_thresholds(), 255 * sizeof(float)); return JP_OK; }
int jp_abi_version(void) { return JP_ABI_VERSION; }
long long jp_gamma_sweep(int n_threads) { return (long long)host_gamma_sweep(n_threads); }
int jp_probe_libm_sincosf(void) { return probe_host_sincosf(); }
int jp_probe_libm_xbsdf(void) { return probe_host_libm(); }

int jp_create_context(int device_id, JpContext** out)
{
	if (!out) return fail(JP_ERR_INVALID_ARGUMENT, "jp_create_context: out is null");
	*out = nullptr;
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(JP_ERR_NO_DEVICE, "jp_create_context: no HIP device visible (this library has no CPU fallback)");
	if (device_id < 0 || device_id >= n) return fail(JP_ERR_NO_DEVICE, "jp_create_context: device id out of range");
	HIP_TRY(hipSetDevice(device_id));
	JpContext* c = new JpContext;
	c->device = device_id;
	std::memset(&c->counters, 0, sizeof(c->counters));
	std::memset(&c->q, 0, sizeof(c->q));
	if (const char* e = getenv("JETPBRT_BLOCKS_PER_CU")) { int v = atoi(e); if (v >= 1 && v <= 256) { c->blocks_per_cu = v; c->bpc_from_env = true; } }
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) c->n_cus = prop.multiProcessorCount;
	if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess
	    || hipMalloc((void**)&c->d_cnt, sizeof(DevCounters)) != hipSuccess)
	{ delete c; return fail(JP_ERR_DEVICE, "jp_create_context: stream/event/counter allocation failed"); }
	c->sincosf_mode = probe_host_sincosf();
	{ hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(jp::g_sincosf_mode), &c->sincosf_mode, sizeof(int)); if (e != hipSuccess) { jp_destroy_context(c); return fail(JP_ERR_DEVICE, std::string("jp_create_context: hipMemcpyToSymbol: ") + hipGetErrorString(e)); } }
	c->libm_mode = probe_host_libm();
	{ hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(jp::xb::g_libm_mode), &c->libm_mode, sizeof(int)); if (e != hipSuccess) { jp_destroy_context(c); return fail(JP_ERR_DEVICE, std::string("jp_create_context: hipMemcpyToSymbol: ") + hipGetErrorString(e)); } }
	*out = c;
	return JP_OK;
}

int jp_destroy_context(JpContext* c)
{
	if (!c) return JP_OK;
	hipSetDevice(c->device);
	if (c->stream) hipStreamSynchronize(c->stream);
	for (JpContext* l : c->lanes) { std::memset(&l->sv, 0, sizeof(l->sv)); jp_destroy_context(l); }
	c->lanes.clear();
	if (c->ev_added) hipEventDestroy(c->ev_added);
	if (!c->is_lane) free_scene(c);
	free_queues(c);
	free_fused(c);
	if (c->d_pix_acc) hipFree(c->d_pix_acc);
	if (c->d_spill) hipFree(c->d_spill);
	if (c->d_spill2) hipFree(c->d_spill2);
	if (c->stream2) { hipStreamSynchronize(c->stream2); hipStreamDestroy(c->stream2); }
	if (c->ev_shade) hipEventDestroy(c->ev_shade); if (c->ev_shadow) hipEventDestroy(c->ev_shadow);
	if (c->d_film) hipFree(c->d_film);
	if (c->h_film) hipHostFree(c->h_film);
	if (c->d_bsdf_in) hipFree(c->d_bsdf_in); if (c->d_bsdf_out) hipFree(c->d_bsdf_out); if (c->d_bsdf_fl) hipFree(c->d_bsdf_fl);
	if (c->d_gamma) hipFree(c->d_gamma);
	if (c->d_rgb8) hipFree(c->d_rgb8);
	if (c->h_rgb8) hipHostFree(c->h_rgb8);
	if (c->d_cnt) hipFree(c->d_cnt);
	for (hipEvent_t e : c->evpool) hipEventDestroy(e);
	if (c->ev0) hipEventDestroy(c->ev0);
	if (c->ev1) hipEventDestroy(c->ev1);
	if (c->stream) hipStreamDestroy(c->stream);
	delete c;
	return JP_OK;
}

} // extern "C"

// ---- scene validation + upload ----------------------------------------------------------------------------------------
namespace
{
struct HV3 { float x, y, z; };
inline HV3 hsub(HV3 a, HV3 b) { HV3 r = { a.x - b.x, a.y - b.y, a.z - b.z }; return r; }
inline HV3 hcross(HV3 a, HV3 v) { HV3 r = { a.y * v.z - a.z * v.y, a.z * v.x - a.x * v.z, a.x * v.y - a.y * v.x }; return r; }
inline float hlen(HV3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
inline HV3 hld(const float* p) { HV3 r = { p[0], p[1], p[2] }; return r; }

// Binned-SAH binary tree over ITEM boxes with one item per leaf (certified walk: the items are the leaves of the caller's tree).
// left[n] >= 0: interior (left[n], right[n]); left[n] < 0: leaf holding item -left[n] - 1.  bounds: 6 floats per node.  Root = node 0.
struct ItemTree { std::vector<int> left, right; std::vector<float> bounds; int height = 0; };
int item_tree_build(const std::vector<float>& ib, std::vector<int>& idx, int start, int end, ItemTree& t, int depth)
{
	const int node = (int)t.left.size(); t.left.push_back(0); t.right.push_back(0); t.bounds.resize(t.bounds.size() + 6);
	t.height = std::max(t.height, depth);
	float nb[6] = { 1e30f, 1e30f, 1e30f, -1e30f, -1e30f, -1e30f }, cb[6] = { 1e30f, 1e30f, 1e30f, -1e30f, -1e30f, -1e30f };
	for (int i = start; i < end; i++)
	{
		const float* b = &ib[6 * (size_t)idx[i]];
		for (int a = 0; a < 3; a++) { nb[a] = std::min(nb[a], b[a]); nb[3 + a] = std::max(nb[3 + a], b[3 + a]); const float c = 0.5f * (b[a] + b[3 + a]); cb[a] = std::min(cb[a], c); cb[3 + a] = std::max(cb[3 + a], c); }
	}
	std::memcpy(&t.bounds[6 * (size_t)node], nb, sizeof(nb));
	if (end - start == 1) { t.left[node] = -idx[start] - 1; return node; }
	auto area = [](const float* b) { const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2]; return (dx < 0 || dy < 0 || dz < 0) ? 0.f : dx * dy + dy * dz + dz * dx; };
	const int NB = 16; float bestCost = 3.0e38f; int bestAxis = -1, bestBin = -1;
	for (int a = 0; a < 3; a++)
	{
		const float lo = cb[a], hi = cb[3 + a]; if (!(hi > lo)) continue;
		float bins[NB][6]; int cnt[NB];
		for (int k = 0; k < NB; k++) { for (int j = 0; j < 3; j++) { bins[k][j] = 1e30f; bins[k][3 + j] = -1e30f; } cnt[k] = 0; }
		const float scale = NB / (hi - lo);
		for (int i = start; i < end; i++)
		{
			const float* b = &ib[6 * (size_t)idx[i]];
			int k = (int)((0.5f * (b[a] + b[3 + a]) - lo) * scale); k = std::max(0, std::min(NB - 1, k));
			for (int j = 0; j < 3; j++) { bins[k][j] = std::min(bins[k][j], b[j]); bins[k][3 + j] = std::max(bins[k][3 + j], b[3 + j]); } cnt[k]++;
		}
		float rightArea[NB]; int rightCnt[NB]; float acc[6] = { 1e30f, 1e30f, 1e30f, -1e30f, -1e30f, -1e30f }; int c = 0;
		for (int k = NB - 1; k > 0; k--) { for (int j = 0; j < 3; j++) { acc[j] = std::min(acc[j], bins[k][j]); acc[3 + j] = std::max(acc[3 + j], bins[k][3 + j]); } c += cnt[k]; rightArea[k] = area(acc); rightCnt[k] = c; }
		for (int j = 0; j < 3; j++) { acc[j] = 1e30f; acc[3 + j] = -1e30f; } c = 0;
		for (int k = 0; k < NB - 1; k++)
		{
			for (int j = 0; j < 3; j++) { acc[j] = std::min(acc[j], bins[k][j]); acc[3 + j] = std::max(acc[3 + j], bins[k][3 + j]); } c += cnt[k];
			if (c == 0 || rightCnt[k + 1] == 0) continue;
			const float cost = area(acc) * c + rightArea[k + 1] * rightCnt[k + 1];
			if (cost < bestCost) { bestCost = cost; bestAxis = a; bestBin = k; }
		}
	}
	int mid = -1;
	if (bestAxis >= 0)
	{
		const int a = bestAxis; const float lo = cb[a], scale = NB / (cb[3 + a] - cb[a]);
		int* m = std::partition(idx.data() + start, idx.data() + end, [&](int i) { const float* b = &ib[6 * (size_t)i]; int k = (int)((0.5f * (b[a] + b[3 + a]) - lo) * scale); k = std::max(0, std::min(NB - 1, k)); return k <= bestBin; });
		mid = (int)(m - idx.data());
	}
	if (mid <= start || mid >= end)
	{   // coinciding centroids: split the range in the middle
		mid = start + (end - start) / 2;
	}
	const int l = item_tree_build(ib, idx, start, mid, t, depth + 1);
	const int r = item_tree_build(ib, idx, mid, end, t, depth + 1);
	t.left[node] = l; t.right[node] = r;
	return node;
}

int bvh_height(const JpScene* s, int node, int depth, int limit, bool& bad, std::vector<char>& seen)
{
	if (node < 0 || node >= s->n_bvh_nodes || seen[node] || depth > limit) { bad = true; return 0; }
	seen[node] = 1;
	if (s->bvh_left[node] < 0) return 0;                               // leaf
	int a = bvh_height(s, s->bvh_left[node], depth + 1, limit, bad, seen);
	int b = bvh_height(s, s->bvh_right[node], depth + 1, limit, bad, seen);
	return 1 + std::max(a, b);
}
}

extern "C" int jp_upload_scene(JpContext* c, const JpScene* s)
{
	if (!c || !s) return fail(JP_ERR_INVALID_ARGUMENT, "jp_upload_scene: null argument");
	// ---- validate every index on the host: a bad index must never reach a kernel ----
	if (s->n_primitives <= 0) return fail(JP_ERR_INVALID_ARGUMENT, "jp_upload_scene: scene has no primitives");
	if (s->n_triangles < 0 || s->n_rectangles < 0 || s->n_spheres < 0 || s->n_disks < 0 || s->n_materials < 0 || s->n_lights < 0 || s->n_bvh_nodes < 0 || s->n_bvh_prim_indices < 0)
		return fail(JP_ERR_INVALID_ARGUMENT, "jp_upload_scene: negative count");
	if (s->bvh_reference_semantics < 0 || s->bvh_reference_semantics > 2) return fail(JP_ERR_INVALID_ARGUMENT, "jp_upload_scene: bvh_reference_semantics must be 0, 1 or 2");
	if (s->bvh_reference_semantics != 0 && s->n_bvh_nodes == 0) return fail(JP_ERR_INVALID_ARGUMENT, "jp_upload_scene: reference semantics need the caller's tree (n_bvh_nodes == 0)");
	const bool device_build = s->n_bvh_nodes == 0;              // no hierarchy handed over: build it on the device (jp_lbvh.h)
	const bool ref_sem = !device_build && (s->bvh_reference_semantics == 1 || s->bvh_reference_semantics == 2);   // walk the caller's tree with the reference's semantics (traverse_ref)
	if (!s->prim_shape_type || !s->prim_shape_index || !s->prim_material || !s->prim_light || (!device_build && (!s->bvh_bounds || !s->bvh_left || !s->bvh_right || !s->bvh_prim_index)))
		return fail(JP_ERR_INVALID_ARGUMENT, "jp_upload_scene: null array");
	if ((s->n_triangles && (!s->tri_p0 || !s->tri_p1 || !s->tri_p2 || !s->tri_n)) || (s->n_rectangles && (!s->rect_p0 || !s->rect_p1 || !s->rect_p2 || !s->rect_p3 || !s->rect_n))
	    || (s->n_spheres && (!s->sph_center || !s->sph_radius)) || (s->n_disks && (!s->disk_center || !s->disk_normal || !s->disk_radius)) || (s->n_materials && (!s->mat_type || !s->mat_params)) || (s->n_lights && (!s->light_type || !s->light_radiance || !s->light_prim)))
		return fail(JP_ERR_INVALID_ARGUMENT, "jp_upload_scene: null array for a non-zero count");
	if (s->n_lights > 255) return fail(JP_ERR_UNSUPPORTED, "jp_upload_scene: more than 255 lights are not supported by the shadow-entry packing");
	bool hasNull = false;
	for (int i = 0; i < s->n_primitives; i++)
	{
		int t = s->prim_shape_type[i], k = s->prim_shape_index[i];
		int lim = t == JP_SHAPE_TRIANGLE ? s->n_triangles : t == JP_SHAPE_RECTANGLE ? s->n_rectangles : t == JP_SHAPE_SPHERE ? s->n_spheres : t == JP_SHAPE_DISK ? s->n_disks : -1;
		if (lim < 0 || k < 0 || k >= lim) return fail(JP_ERR_INVALID_ARGUMENT, "jp_upload_scene: primitive shape reference out of range");
		if (s->prim_material[i] < -1 || s->prim_material[i] >= s->n_materials) return fail(JP_ERR_INVALID_ARGUMENT, "jp_upload_scene: primitive material out of range");
		if (s->prim_light[i] < -1 || s->prim_light[i] >= s->n_lights) return fail(JP_ERR_INVALID_ARGUMENT, "jp_upload_scene: primitive light out of range");
		if (s->prim_light[i] >= 0 && s->light_type[s->prim_light[i]] != JP_LIGHT_AREA) return fail(JP_ERR_INVALID_ARGUMENT, "jp_upload_scene: primitive light is not an area light");
		if (s->prim_material[i] < 0) hasNull = true;
	}
	for (int i = 0; i < s->n_materials; i++) if (s->mat_type[i] < JP_MAT_MATTE || s->mat_type[i] > JP_MAT_METAL) return fail(JP_ERR_INVALID_ARGUMENT, "jp_upload_scene: unknown material type");
	for (int i = 0; i < s->n_lights; i++)
	{
		if (s->light_type[i] == JP_LIGHT_AREA) { if (s->light_prim[i] < 0 || s->light_prim[i] >= s->n_primitives) return fail(JP_ERR_INVALID_ARGUMENT, "jp_upload_scene: area light primitive out of range"); }
		else if (s->light_type[i] == JP_LIGHT_POINT || s->light_type[i] == JP_LIGHT_DIRECTION) { if (!s->light_vec) return fail(JP_ERR_INVALID_ARGUMENT, "jp_upload_scene: point / direction light without light_vec"); }
		else if (s->light_type[i] != JP_LIGHT_ENVIRONMENT) return fail(JP_ERR_INVALID_ARGUMENT, "jp_upload_scene: unknown light type");
	}
	// BVH: a tree, every primitive in exactly one leaf, leaf ranges in bounds, height within the LDS stack
	std::vector<char> seen(s->n_bvh_nodes, 0); bool bad = false;
	int height = device_build ? 0 : bvh_height(s, 0, 0, 4 * JP_STACK_DEPTH, bad, seen);
	if (bad) return fail(JP_ERR_INVALID_ARGUMENT, "jp_upload_scene: BVH is not a tree rooted at node 0 (cycle, bad child index or excessive depth)");
	if (!device_build && height + 1 > JP_STACK_DEPTH) return fail(JP_ERR_INVALID_ARGUMENT, "jp_upload_scene: BVH height exceeds the device traversal stack (32)");
	std::vector<int> primSeen(s->n_primitives, 0);
	for (int n = 0; n < s->n_bvh_nodes; n++)
	{
		if (!seen[n]) continue;
		if (s->bvh_left[n] >= 0) continue;
		int first = -s->bvh_left[n] - 1, cnt = s->bvh_right[n];
		if (cnt < 1 || cnt > 16 || first < 0 || first + cnt > s->n_bvh_prim_indices) return fail(JP_ERR_INVALID_ARGUMENT, "jp_upload_scene: BVH leaf range invalid (1..16 primitives per leaf)");
		for (int k = 0; k < cnt; k++) { int p = s->bvh_prim_index[first + k]; if (p < 0 || p >= s->n_primitives) return fail(JP_ERR_INVALID_ARGUMENT, "jp_upload_scene: BVH primitive index out of range"); primSeen[p]++; }
	}
	if (!device_build) for (int i = 0; i < s->n_primitives; i++) if (primSeen[i] != 1) return fail(JP_ERR_INVALID_ARGUMENT, "jp_upload_scene: every primitive must be in exactly one BVH leaf");

	HIP_TRY(hipSetDevice(c->device));
	HIP_TRY(hipStreamSynchronize(c->stream));
	free_scene(c);

	// ---- device primitive records in leaf order + device BVH (children boxes in the parent) ----
	std::vector<int> hostToDevNode(s->n_bvh_nodes, -1), devPrimOf(s->n_primitives, -1);
	std::vector<float4> nodes; std::vector<float4> prims; std::vector<int4> meta;
	// JETPBRT_BOX_PAD (diagnosis only, tools/gpu_fringe_census.py): every box of the host-built trees grows by this many scene units, so the walk
	// also visits the leaves whose triangles accept a hit in the fp32 fringe OUTSIDE their exact box -- a stand-in for testing every primitive
	float extra_pad = 0.f; if (const char* ev = getenv("JETPBRT_BOX_PAD")) extra_pad = std::max(0.f, (float)atof(ev));
	auto pad_box = [&](int n, float* b) {
		for (int a = 0; a < 3; a++)
		{
			float lo = s->bvh_bounds[6 * n + a], hi = s->bvh_bounds[6 * n + 3 + a];
			float m = std::max(std::fabs(lo), std::fabs(hi)); float e = m * 1e-6f + 1e-6f + extra_pad;   // >> ulp(m): flat (zero-extent) boxes stay hittable
			b[a] = lo - e; b[3 + a] = hi + e;
		}
	};
	auto emit_prim = [&](int p) -> int {
		const int dev = (int)meta.size(); devPrimOf[p] = dev;
		int t = s->prim_shape_type[p], i = s->prim_shape_index[p];
		float4 g[4] = { make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0) };
		if (t == JP_SHAPE_TRIANGLE)
		{
			g[0] = make_float4(s->tri_p0[3 * i], s->tri_p0[3 * i + 1], s->tri_p0[3 * i + 2], 0); g[1] = make_float4(s->tri_p1[3 * i], s->tri_p1[3 * i + 1], s->tri_p1[3 * i + 2], 0);
			g[2] = make_float4(s->tri_p2[3 * i], s->tri_p2[3 * i + 1], s->tri_p2[3 * i + 2], 0); g[3] = make_float4(s->tri_n[3 * i], s->tri_n[3 * i + 1], s->tri_n[3 * i + 2], 0);
		}
		else if (t == JP_SHAPE_RECTANGLE)
		{
			g[0] = make_float4(s->rect_p0[3 * i], s->rect_p0[3 * i + 1], s->rect_p0[3 * i + 2], s->rect_p3[3 * i]);
			g[1] = make_float4(s->rect_p1[3 * i], s->rect_p1[3 * i + 1], s->rect_p1[3 * i + 2], s->rect_p3[3 * i + 1]);
			g[2] = make_float4(s->rect_p2[3 * i], s->rect_p2[3 * i + 1], s->rect_p2[3 * i + 2], s->rect_p3[3 * i + 2]);
			g[3] = make_float4(s->rect_n[3 * i], s->rect_n[3 * i + 1], s->rect_n[3 * i + 2], 0);
		}
		else if (t == JP_SHAPE_DISK)
		{
			g[0] = make_float4(s->disk_center[3 * i], s->disk_center[3 * i + 1], s->disk_center[3 * i + 2], s->disk_radius[i]);
			g[1] = make_float4(s->disk_normal[3 * i], s->disk_normal[3 * i + 1], s->disk_normal[3 * i + 2], 0);
		}
		else g[0] = make_float4(s->sph_center[3 * i], s->sph_center[3 * i + 1], s->sph_center[3 * i + 2], s->sph_radius[i]);
		int tb = t; std::memcpy(&g[3].w, &tb, 4);
		for (int j = 0; j < 4; j++) prims.push_back(g[j]);
		int4 m; m.x = p; m.y = s->prim_material[p]; m.z = s->prim_light[p]; m.w = t; meta.push_back(m);
		return dev;
	};
	auto emit_leaf = [&](int n) -> int {
		int first = -s->bvh_left[n] - 1, cnt = s->bvh_right[n];
		int dfirst = devPrimOf[s->bvh_prim_index[first]];                    // already placed by the wide-tree pass?
		if (dfirst < 0) { dfirst = (int)meta.size(); for (int k = 0; k < cnt; k++) emit_prim(s->bvh_prim_index[first + k]); }
		return -(((dfirst << 4) | (cnt - 1)) + 1);
	};

	// ---- large scenes: collapse the binary tree into 8-wide nodes with quantised child boxes (traverse_wide) ----
	int nleaves_total = 0; for (int n = 0; n < s->n_bvh_nodes; n++) if (seen[n] && s->bvh_left[n] < 0) nleaves_total++;
	std::vector<uint32_t> wide; int wide_height = 0;
	bool use_wide = !device_build && !ref_sem && nleaves_total > 32 && s->bvh_left[0] >= 0;
	if (!device_build && !ref_sem)
	{
		size_t est = ((size_t)s->n_bvh_nodes + (size_t)s->n_primitives) * 80;                            // LDS-resident scenes keep the binary tree
		if (est + (size_t)(height + 2) * JP_BLOCK * sizeof(int) <= 40 * 1024) use_wide = false;
		if (const char* e = getenv("JETPBRT_TRAVERSAL")) { int m = atoi(e); if (m == 3 && s->bvh_left[0] >= 0) use_wide = true; else if (m >= 0 && m <= 2) use_wide = false; }
	}
	if (use_wide)
	{
		struct Child { int node; int first, cnt, leaf_first, leaf_cnt; float b[6]; };   // node >= 0: inner (binary node index); else a chunk of <= 3 primitives of one binary leaf
		struct Item { int bnode; uint32_t widx; int depth; };
		auto area = [](const float* b) { float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2]; return dx * dy + dy * dz + dz * dx; };
		std::vector<Item> queue; queue.push_back({ 0, 0u, 1 });
		wide.assign(20, 0u);
		bool ok = true;
		std::vector<Child> ch; ch.reserve(16);                   // scratch reused across nodes (no allocation per wide node)
		queue.reserve((size_t)s->n_bvh_nodes / 2 + 16); wide.reserve(((size_t)s->n_bvh_nodes / 2 + 16) * 20);
		for (size_t qi = 0; qi < queue.size() && ok; qi++)
		{
			const Item it = queue[qi];
			wide_height = std::max(wide_height, it.depth);
			// gather up to 8 child slots: open the inner child with the largest box while the slots allow it
			ch.clear();
			auto add = [&](int n) {
				float b[6]; pad_box(n, b);
				if (s->bvh_left[n] >= 0) { Child c; c.node = n; c.first = c.cnt = c.leaf_first = c.leaf_cnt = 0; std::memcpy(c.b, b, sizeof(b)); ch.push_back(c); }
				else
				{
					int first = -s->bvh_left[n] - 1, cnt = s->bvh_right[n];
					for (int k = 0; k < cnt; k += 3) { Child c; c.node = -1; c.first = first + k; c.cnt = std::min(3, cnt - k); c.leaf_first = first; c.leaf_cnt = cnt; std::memcpy(c.b, b, sizeof(b)); ch.push_back(c); }
				}
			};
			auto slots_of = [&](int n) { return s->bvh_left[n] >= 0 ? 1 : (s->bvh_right[n] + 2) / 3; };
			add(s->bvh_left[it.bnode]); add(s->bvh_right[it.bnode]);
			for (;;)
			{
				int best = -1; float bestA = -1.f;
				for (size_t k = 0; k < ch.size(); k++)
					if (ch[k].node >= 0)
					{
						int need = (int)ch.size() - 1 + slots_of(s->bvh_left[ch[k].node]) + slots_of(s->bvh_right[ch[k].node]);
						if (need <= 8 && area(ch[k].b) > bestA) { bestA = area(ch[k].b); best = (int)k; }
					}
				if (best < 0) break;
				const int n = ch[best].node; ch.erase(ch.begin() + best);
				add(s->bvh_left[n]); add(s->bvh_right[n]);
			}
			if (ch.size() > 8) { ok = false; break; }
			// node box, scale exponents
			float lo[3] = { 1e30f, 1e30f, 1e30f }, hi[3] = { -1e30f, -1e30f, -1e30f };
			for (const Child& c : ch) for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], c.b[a]); hi[a] = std::max(hi[a], c.b[3 + a]); }
			int eb[3]; float sc3[3];
			for (int a = 0; a < 3; a++)
			{
				int e = (int)std::ceil(std::log2(std::max((hi[a] - lo[a]) / 255.f, 1e-30f)));
				e = std::max(-120, std::min(120, e));
				eb[a] = e + 127; sc3[a] = std::ldexp(1.0f, e);
			}
			// slots: the three bits of a slot say on which side of the node centre the child lies (greedy assignment)
			int slotOf[8]; bool used[8] = { false, false, false, false, false, false, false, false };
			{
				struct Cand { float score; int child, slot; };
				Cand cands[64]; int ncand = 0;                     // <= 8 children x 8 slots, on the stack
				for (size_t k = 0; k < ch.size(); k++) for (int sl = 0; sl < 8; sl++)
				{
					float sc = 0;
					for (int a = 0; a < 3; a++) { float cc = 0.5f * (ch[k].b[a] + ch[k].b[3 + a]) - 0.5f * (lo[a] + hi[a]); sc += ((sl >> a) & 1) ? cc : -cc; }
					cands[ncand++] = { sc, (int)k, sl };
				}
				std::sort(cands, cands + ncand, [](const Cand& x, const Cand& y) { return x.score > y.score; });
				int got[8] = { -1, -1, -1, -1, -1, -1, -1, -1 };
				for (int ci = 0; ci < ncand; ci++) { const Cand& cd = cands[ci]; if (got[cd.child] < 0 && !used[cd.slot]) { got[cd.child] = cd.slot; used[cd.slot] = true; } }
				for (size_t k = 0; k < ch.size(); k++) slotOf[k] = got[k];
			}
			// emit: inner children get consecutive wide indices in slot order; leaf chunks append their primitives
			uint8_t metaB[8] = { 0 }, ql[3][8], qh[3][8]; uint32_t imask = 0;
			for (int sl = 0; sl < 8; sl++) for (int a = 0; a < 3; a++) { ql[a][sl] = 255; qh[a][sl] = 0; }
			const uint32_t child_base = (uint32_t)(wide.size() / 20);
			const uint32_t prim_base = (uint32_t)meta.size();
			int order[8], no = 0; for (int sl = 0; sl < 8; sl++) for (size_t k = 0; k < ch.size(); k++) if (slotOf[k] == sl) order[no++] = (int)k;
			uint32_t ninner = 0; int poff = 0;
			for (int oi = 0; oi < no; oi++)
			{
				const Child& c = ch[order[oi]]; const int sl = slotOf[order[oi]];
				if (c.node >= 0) { imask |= 1u << sl; metaB[sl] = (uint8_t)(0x20 | (24 + sl)); queue.push_back({ c.node, child_base + ninner, it.depth + 1 }); ninner++; wide.resize(wide.size() + 20, 0u); }
				else
				{
					// the whole binary leaf is emitted when its first chunk comes up, so that its primitives stay contiguous on the
					// device and the binary tree (used for closest-hit rays) can address the same records
					if (devPrimOf[s->bvh_prim_index[c.first]] < 0) for (int k = 0; k < c.leaf_cnt; k++) emit_prim(s->bvh_prim_index[c.leaf_first + k]);
					poff = devPrimOf[s->bvh_prim_index[c.first]] - (int)prim_base;
					if (poff < 0 || poff + c.cnt > 24) { ok = false; break; }
					metaB[sl] = (uint8_t)((((1u << c.cnt) - 1u) << 5) | (unsigned)poff);
				}
				for (int a = 0; a < 3; a++)
				{
					int q0 = (int)std::floor((c.b[a] - lo[a]) / sc3[a]), q1 = (int)std::ceil((c.b[3 + a] - lo[a]) / sc3[a]);
					q0 = std::max(0, std::min(255, q0)); q1 = std::max(0, std::min(255, q1));
					while (q0 > 0 && std::fmaf((float)q0, sc3[a], lo[a]) > c.b[a]) q0--;                     // conservative in fp32, as the device evaluates it
					while (q1 < 255 && std::fmaf((float)q1, sc3[a], lo[a]) < c.b[3 + a]) q1++;
					if (std::fmaf((float)q1, sc3[a], lo[a]) < c.b[3 + a]) { ok = false; break; }
					ql[a][sl] = (uint8_t)q0; qh[a][sl] = (uint8_t)q1;
				}
				if (!ok) break;
			}
			if (!ok) break;
			auto pack4 = [](const uint8_t* v) { return (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24); };
			uint32_t* w = &wide[(size_t)it.widx * 20];
			std::memcpy(&w[0], &lo[0], 4); std::memcpy(&w[1], &lo[1], 4); std::memcpy(&w[2], &lo[2], 4);
			w[3] = (uint32_t)eb[0] | ((uint32_t)eb[1] << 8) | ((uint32_t)eb[2] << 16) | (imask << 24);
			w[4] = child_base; w[5] = prim_base; w[6] = pack4(metaB); w[7] = pack4(metaB + 4);
			w[8] = pack4(ql[0]); w[9] = pack4(ql[0] + 4); w[10] = pack4(ql[1]); w[11] = pack4(ql[1] + 4);
			w[12] = pack4(ql[2]); w[13] = pack4(ql[2] + 4); w[14] = pack4(qh[0]); w[15] = pack4(qh[0] + 4);
			w[16] = pack4(qh[1]); w[17] = pack4(qh[1] + 4); w[18] = pack4(qh[2]); w[19] = pack4(qh[2] + 4);
		}
		if (!ok || (int)meta.size() != s->n_primitives)
		{   // a foreign BVH with leaves too large for the wide layout: keep the binary tree
			use_wide = false; wide.clear(); prims.clear(); meta.clear(); std::fill(devPrimOf.begin(), devPrimOf.end(), -1);
		}
	}

	// binary device tree (small and medium scenes): interior nodes get device indices in DFS order
	std::vector<int> order;
	if (!device_build && !ref_sem) { std::vector<int> st; st.push_back(0); while (!st.empty()) { int n = st.back(); st.pop_back(); if (s->bvh_left[n] < 0) continue; hostToDevNode[n] = (int)order.size(); order.push_back(n); st.push_back(s->bvh_right[n]); st.push_back(s->bvh_left[n]); } }
	const float kEmpty[6] = { 1e30f, 1e30f, 1e30f, -1e30f, -1e30f, -1e30f };
	std::vector<int> cert_item_first, cert_item_cnt; std::vector<float> cert_item_box;     // reference semantics: the leaves of the caller's tree (device primitive range, exact box)
	if (device_build) {}
	else if (ref_sem)
	{   // the caller's nodes under their own indices, unpadded boxes; primitives in the leaves' visiting order (left before right)
		nodes.assign((size_t)2 * s->n_bvh_nodes, make_float4(0, 0, 0, 0));
		std::vector<int> st; st.push_back(0);
		while (!st.empty())
		{
			const int n = st.back(); st.pop_back();
			const float* b = s->bvh_bounds + 6 * (size_t)n;
			int l = s->bvh_left[n], r = s->bvh_right[n];
			if (l < 0)
			{
				const int first = -l - 1, cnt = r;
				const int dfirst = (int)meta.size();
				for (int k = 0; k < cnt; k++) emit_prim(s->bvh_prim_index[first + k]);
				l = -dfirst - 1;
				cert_item_first.push_back(dfirst); cert_item_cnt.push_back(cnt); cert_item_box.insert(cert_item_box.end(), b, b + 6);
			}
			else { st.push_back(r); st.push_back(l); }
			float fl, fr; std::memcpy(&fl, &l, 4); std::memcpy(&fr, &r, 4);
			nodes[2 * (size_t)n] = make_float4(b[0], b[1], b[2], fl); nodes[2 * (size_t)n + 1] = make_float4(b[3], b[4], b[5], fr);
		}
	}
	else if (order.empty())
	{   // the root itself is a leaf: a synthetic interior root whose right child can never be hit
		float lb[6]; pad_box(0, lb);
		int ref = emit_leaf(0), rr = ref; float fr, fl; std::memcpy(&fl, &ref, 4); std::memcpy(&fr, &rr, 4);
		nodes.push_back(make_float4(lb[0], lb[1], lb[2], lb[3])); nodes.push_back(make_float4(lb[4], lb[5], kEmpty[0], kEmpty[1]));
		nodes.push_back(make_float4(kEmpty[2], kEmpty[3], kEmpty[4], kEmpty[5])); nodes.push_back(make_float4(fl, fr, 0, 0));
	}
	else
	{
		nodes.resize(4 * order.size());
		for (size_t di = 0; di < order.size(); di++)
		{
			int n = order[di], l = s->bvh_left[n], r = s->bvh_right[n];
			float lb[6], rb[6]; pad_box(l, lb); pad_box(r, rb);
			int lref = s->bvh_left[l] < 0 ? emit_leaf(l) : hostToDevNode[l];
			int rref = s->bvh_left[r] < 0 ? emit_leaf(r) : hostToDevNode[r];
			float fl, fr; std::memcpy(&fl, &lref, 4); std::memcpy(&fr, &rref, 4);
			nodes[4 * di + 0] = make_float4(lb[0], lb[1], lb[2], lb[3]); nodes[4 * di + 1] = make_float4(lb[4], lb[5], rb[0], rb[1]);
			nodes[4 * di + 2] = make_float4(rb[2], rb[3], rb[4], rb[5]); nodes[4 * di + 3] = make_float4(fl, fr, 0, 0);
		}
	}
	if ((size_t)s->n_primitives >= (1u << 27)) return fail(JP_ERR_UNSUPPORTED, "jp_upload_scene: too many primitives for the leaf reference encoding");

	// ---- large scenes: the binary tree collapsed into 4-wide nodes with quantised child boxes for the closest-hit rays (Walker<4>) ----
	// From binary node b: its two children, then the interior child with the largest box is opened again while fewer than four
	// slots are taken.  Leaves keep the binary tree's encoding and primitive records.  JETPBRT_Q4=0: closest hits walk the binary tree.
	// (one collapse for two sources: the caller's tree as it is, and -- reference semantics, certified walk -- the tree built below over the caller's leaves)
	auto collapse_q4 = [&](size_t n_nodes, auto isInner, auto leftOf, auto rightOf, auto boxOf, auto leafRefOf, auto flagOf, std::vector<uint32_t>& q4, int& q4_height) -> bool
	{
		struct Item { int bnode; uint32_t idx; int depth; };
		auto area = [](const float* b) { float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2]; return dx * dy + dy * dz + dz * dx; };
		std::vector<Item> queue; queue.reserve(n_nodes / 2 + 16); queue.push_back({ 0, 0u, 1 });
		q4.clear(); q4_height = 0; q4.reserve((n_nodes / 2 + 16) * 16); q4.assign(16, 0u);
		bool ok = true;
		for (size_t qi = 0; qi < queue.size() && ok; qi++)
		{
			const Item it = queue[qi];
			q4_height = std::max(q4_height, it.depth);
			int ch[4]; float cb[4][6]; int nc = 0;
			ch[nc] = leftOf(it.bnode); boxOf(ch[nc], cb[nc]); nc++;
			ch[nc] = rightOf(it.bnode); boxOf(ch[nc], cb[nc]); nc++;
			while (nc < 4)
			{
				int best = -1; float bestA = -1.f;
				for (int k = 0; k < nc; k++) if (isInner(ch[k]) && area(cb[k]) > bestA) { bestA = area(cb[k]); best = k; }
				if (best < 0) break;
				const int n = ch[best];
				ch[best] = leftOf(n); boxOf(ch[best], cb[best]);
				ch[nc] = rightOf(n); boxOf(ch[nc], cb[nc]); nc++;
			}
			float lo[3] = { 1e30f, 1e30f, 1e30f }, hi[3] = { -1e30f, -1e30f, -1e30f };
			for (int k = 0; k < nc; k++) for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], cb[k][a]); hi[a] = std::max(hi[a], cb[k][3 + a]); }
			int eb[3]; float sc3[3];
			for (int a = 0; a < 3; a++)
			{
				int e = (int)std::ceil(std::log2(std::max((hi[a] - lo[a]) / 255.f, 1e-30f)));
				e = std::max(-120, std::min(120, e));
				eb[a] = e + 127; sc3[a] = std::ldexp(1.0f, e);
			}
			// Walker<4> evaluates a slab distance as q * (2^e / d) + (p - o) / d: its rounding error grows with the NODE's extent, so every
			// child box gets 1e-6 of the node's extent on top of the relative padding of the box source before it is quantised outward
			for (int k = 0; k < nc; k++) for (int a = 0; a < 3; a++) { const float ex = 1e-6f * (hi[a] - lo[a]); cb[k][a] -= ex; cb[k][3 + a] += ex; }
			for (int a = 0; a < 3; a++) { const float ex = 1e-6f * (hi[a] - lo[a]); lo[a] -= ex; hi[a] += ex; }
			for (int a = 0; a < 3; a++)
			{
				int e = (int)std::ceil(std::log2(std::max((hi[a] - lo[a]) / 255.f, 1e-30f)));
				e = std::max(-120, std::min(120, e));
				eb[a] = e + 127; sc3[a] = std::ldexp(1.0f, e);
			}
			uint8_t ql[3][4], qh[3][4]; uint32_t refs[4] = { 0, 0, 0, 0 }, valid = 0, flags = 0;
			for (int k = 0; k < 4; k++) for (int a = 0; a < 3; a++) { ql[a][k] = 255; qh[a][k] = 0; }
			for (int k = 0; k < nc && ok; k++)
			{
				valid |= 1u << k;
				const int n = ch[k];
				if (flagOf(n)) flags |= 1u << k;
				int r;
				if (isInner(n)) { r = (int)(q4.size() / 16); queue.push_back({ n, (uint32_t)r, it.depth + 1 }); q4.resize(q4.size() + 16, 0u); }
				else r = leafRefOf(n);
				std::memcpy(&refs[k], &r, 4);
				for (int a = 0; a < 3; a++)
				{
					int q0 = (int)std::floor((cb[k][a] - lo[a]) / sc3[a]), q1 = (int)std::ceil((cb[k][3 + a] - lo[a]) / sc3[a]);
					q0 = std::max(0, std::min(255, q0)); q1 = std::max(0, std::min(255, q1));
					while (q0 > 0 && std::fmaf((float)q0, sc3[a], lo[a]) > cb[k][a]) q0--;                     // conservative in fp32, as the device evaluates it
					while (q1 < 255 && std::fmaf((float)q1, sc3[a], lo[a]) < cb[k][3 + a]) q1++;
					if (std::fmaf((float)q1, sc3[a], lo[a]) < cb[k][3 + a] || std::fmaf((float)q0, sc3[a], lo[a]) > cb[k][a]) { ok = false; break; }
					ql[a][k] = (uint8_t)q0; qh[a][k] = (uint8_t)q1;
				}
			}
			if (!ok) break;
			auto pack4 = [](const uint8_t* v) { return (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24); };
			uint32_t* w = &q4[(size_t)it.idx * 16];
			std::memcpy(&w[0], &lo[0], 4); std::memcpy(&w[1], &lo[1], 4); std::memcpy(&w[2], &lo[2], 4);
			w[3] = (uint32_t)eb[0] | ((uint32_t)eb[1] << 8) | ((uint32_t)eb[2] << 16) | (valid << 24);
			w[4] = refs[0]; w[5] = refs[1]; w[6] = refs[2]; w[7] = refs[3];
			w[8] = pack4(ql[0]); w[9] = pack4(ql[1]); w[10] = pack4(ql[2]); w[11] = pack4(qh[0]);
			w[12] = pack4(qh[1]); w[13] = pack4(qh[2]); w[14] = flags; w[15] = 0;
		}
		return ok;
	};
	std::vector<uint32_t> q4; int q4_height = 0;
	bool use_q4 = !device_build && !ref_sem && s->n_primitives > 1024 && s->bvh_left[0] >= 0 && !order.empty();
	if (const char* e = getenv("JETPBRT_Q4")) use_q4 = use_q4 && atoi(e) != 0;
	if (use_q4)
	{
		const bool ok = collapse_q4((size_t)s->n_bvh_nodes, [&](int n) { return s->bvh_left[n] >= 0; }, [&](int n) { return s->bvh_left[n]; }, [&](int n) { return s->bvh_right[n]; },
		                            [&](int n, float* bb) { pad_box(n, bb); }, [&](int n) { return emit_leaf(n); }, [](int) { return false; }, q4, q4_height);
		if (!ok || (int)meta.size() != s->n_primitives) { use_q4 = false; q4.clear(); }
	}

	// ---- reference semantics on large scenes: the certified walk (Walker<6>, jp_device.h) ----
	// A binned-SAH tree over the LEAVES of the caller's tree (their exact boxes, padded like every box of the ordered walks), collapsed to 4-wide
	// nodes; a leaf of it is one leaf of the caller's tree (same primitive range, same order).  Per primitive the exact box of its leaf
	// (the certificate is FBounds3::Intersect on that box).  The caller's nodes stay on the device for the rays that get no certificate.
	bool want_cert = s->bvh_reference_semantics == 2;
	if (const char* e = getenv("JETPBRT_CERTIFIED")) want_cert = atoi(e) != 0;                // (experiments: either way round)
	bool use_cert = want_cert && ref_sem && s->n_primitives > 1024 && cert_item_first.size() >= 64;
	std::vector<float4> refbox; float cert_pad = 0.f, cert_pad_eye = 0.f;
	if (use_cert)
	{
		const int ni = (int)cert_item_first.size();
		ItemTree it; std::vector<int> idx(ni); for (int i = 0; i < ni; i++) idx[i] = i;
		it.left.reserve(2 * (size_t)ni); it.right.reserve(2 * (size_t)ni); it.bounds.reserve(12 * (size_t)ni);
		item_tree_build(cert_item_box, idx, 0, ni, it, 1);
		auto box_of = [&](int n, float* bb) {
			for (int a = 0; a < 3; a++)
			{
				const float lo = it.bounds[6 * (size_t)n + a], hi = it.bounds[6 * (size_t)n + 3 + a];
				const float m = std::max(std::fabs(lo), std::fabs(hi)), e = m * 1e-6f + 1e-6f + extra_pad;
				bb[a] = lo - e; bb[3 + a] = hi + e;
			}
		};
		bool ok = it.left[0] >= 0 && it.height + 2 <= 48;
		for (int i = 0; i < ni && ok; i++) if (cert_item_cnt[i] < 1 || cert_item_cnt[i] > 16) ok = false;
		// "edge-on to the camera": a leaf holding a flat primitive whose plane passes the eye within tau of its distance -- the only primitives a CAMERA ray can
		// lie in to within fp32 noise, i.e. whose acceptance far in front of their leaf's box an ordered walk would cull (Walker<6>).  Flag = leaf, and every node above it.
		float tau = 5e-3f; if (const char* e = getenv("JETPBRT_CERT_EYE")) tau = std::max(0.f, (float)atof(e));
		std::vector<char> item_eye(ni, 0), node_eye(it.left.size(), 0); int n_eye = 0;
		for (int i = 0; i < ni && ok; i++)
			for (int k = 0; k < cert_item_cnt[i]; k++)
			{
				const size_t p = (size_t)cert_item_first[i] + k;
				int type; std::memcpy(&type, &prims[4 * p + 3].w, 4);
				if (type == JP_SHAPE_SPHERE) continue;
				const float4 g0 = prims[4 * p], gn = type == JP_SHAPE_DISK ? prims[4 * p + 1] : prims[4 * p + 3];
				const double vx = (double)g0.x - s->camera.pos[0], vy = (double)g0.y - s->camera.pos[1], vz = (double)g0.z - s->camera.pos[2];
				const double nl = std::sqrt((double)gn.x * gn.x + (double)gn.y * gn.y + (double)gn.z * gn.z), dist = std::sqrt(vx * vx + vy * vy + vz * vz);
				if (std::fabs(vx * gn.x + vy * gn.y + vz * gn.z) <= tau * dist * nl + 1e-30) { if (!item_eye[i]) n_eye++; item_eye[i] = 1; }
			}
		if (ok) for (size_t n = it.left.size(); n-- > 0;) node_eye[n] = it.left[n] < 0 ? item_eye[-it.left[n] - 1] : (char)(node_eye[it.left[n]] | node_eye[it.right[n]]);   // children have higher indices than their parent
		c->cert_eye_leaves = n_eye;
		if (ok) ok = collapse_q4(it.left.size(), [&](int n) { return it.left[n] >= 0; }, [&](int n) { return it.left[n]; }, [&](int n) { return it.right[n]; }, box_of,
		                         [&](int n) { const int item = -it.left[n] - 1; return -(((cert_item_first[item] << 4) | (cert_item_cnt[item] - 1)) + 1); }, [&](int n) { return node_eye[n] != 0; }, q4, q4_height);
		if (!ok) { use_cert = false; q4.clear(); }
		else
		{
			refbox.resize((size_t)2 * s->n_primitives);
			double diag = 0;
			for (int i = 0; i < ni; i++)
			{
				const float* b = &cert_item_box[6 * (size_t)i];
				for (int k = 0; k < cert_item_cnt[i]; k++) { const size_t p = (size_t)cert_item_first[i] + k; refbox[2 * p] = make_float4(b[0], b[1], b[2], 0.f); refbox[2 * p + 1] = make_float4(b[3], b[4], b[5], 0.f); }
				diag += std::sqrt((double)(b[3] - b[0]) * (b[3] - b[0]) + (double)(b[4] - b[1]) * (b[4] - b[1]) + (double)(b[5] - b[2]) * (b[5] - b[2]));
			}
			// distance-cull slack: a hit in the fp32 acceptance fringe of FTriangle::Intersect lies up to ~ eps * D^2 / edge beside its triangle (D: distance
			// from the ray origin), so up to a few times that in front of its leaf's box -- with a 1 / distance tail for rays grazing the box: tmax + K * eps / (mean leaf
			// diagonal) * tmax^2.  K = 1024: 3 of 259,200 pixels of the configs[4] shard (3.1e9 rays) off; 16384: none, for 4 % of the frame rate (profiles/r03l_certified_walk.txt)
			float K = 16384.f; if (const char* e = getenv("JETPBRT_CERT_SLACK")) K = std::max(0.f, (float)atof(e));
			cert_pad = (float)(K * 1.1920929e-7 / std::max(1e-20, diag / ni));
			// rays from the camera position: their noise planes are covered by the edge-on flags, so the slack only has to cover the fringe in front of a leaf's box
			float Ke = std::min(K, 1024.f); if (const char* e = getenv("JETPBRT_CERT_SLACK_EYE")) Ke = std::max(0.f, (float)atof(e));
			cert_pad_eye = (float)(Ke * 1.1920929e-7 / std::max(1e-20, diag / ni));
		}
	}

	// ---- no hierarchy handed over: records go up in creation order and the tree is built on the device (jp_lbvh.h) ----
	size_t n4nodes = nodes.size(), n4prims = prims.size(), nmeta = meta.size();
	bool dev_wide = false; int dev_n_wide = 0; bool dev_q4 = false; int dev_n_q4 = 0;
	c->build_on_device = device_build; c->build_ms = 0.f;
	if (device_build)
	{
		for (int p = 0; p < s->n_primitives; p++) emit_prim(p);
		void *d_p0 = nullptr, *d_m0 = nullptr;
		hipError_t e = hipMalloc(&d_p0, prims.size() * sizeof(float4)); if (e == hipSuccess) e = hipMalloc(&d_m0, meta.size() * sizeof(int4));
		if (e == hipSuccess) e = hipMemcpyAsync(d_p0, prims.data(), prims.size() * sizeof(float4), hipMemcpyHostToDevice, c->stream);
		if (e == hipSuccess) e = hipMemcpyAsync(d_m0, meta.data(), meta.size() * sizeof(int4), hipMemcpyHostToDevice, c->stream);
		LbvhResult lr; std::vector<int> sorted;
		// leaf size: LBVH 3 (round 1, 280k-triangle scene: 540 / 578 / 579 / 560 / 534 / 497 Msamples/s for 1 / 2 / 3 / 4 / 6 / 8); PLOC 2 (round 3, with the 4-wide
		// tree: k_extend 28.0 / 25.3 / 26.1 / 27.1 ms and k_shadow 21.7 / 19.8 / 20.5 / 21.6 ms per 256 spp for 1 / 2 / 3 / 4, profiles/r03g_ploc_ab.txt)
		bool ploc = true;
		if (const char* ev = getenv("JETPBRT_DEVICE_TREE")) ploc = std::string(ev) != "lbvh";
		int maxLeaf = ploc ? 2 : 3;
		if (const char* ev = getenv("JETPBRT_BVH_MAXLEAF")) { int v = atoi(ev); if (v >= 1 && v <= 16) maxLeaf = v; }
		// [round 3] PLOC clustering (jp_ploc.h) instead of the Karras topology; JETPBRT_DEVICE_TREE=lbvh restores the latter, which also serves
		// as the fallback should the clustering not finish within its round limit
		if (e == hipSuccess && ploc) { e = ploc_build(c->stream, (const float4*)d_p0, (const int4*)d_m0, s->n_primitives, maxLeaf, lr, sorted); if (e == hipErrorNotReady) { e = hipSuccess; ploc = false; } }
		if (e == hipSuccess && !ploc) e = lbvh_build(c->stream, (const float4*)d_p0, (const int4*)d_m0, s->n_primitives, maxLeaf, lr, sorted);
		if (d_p0) hipFree(d_p0); if (d_m0) hipFree(d_m0);
		if (e != hipSuccess) return fail(JP_ERR_DEVICE, std::string("jp_upload_scene: device BVH build failed: ") + hipGetErrorString(e));
		if (lr.height + 2 > 60)
		{
			hipFree(lr.d_nodes); hipFree(lr.d_prims); hipFree(lr.d_meta);
			return fail(JP_ERR_UNSUPPORTED, "jp_upload_scene: device-built BVH is deeper than the 58-entry traversal stack; hand over a host-built hierarchy for this scene");
		}
		c->d_nodes = lr.d_nodes; c->d_prims = lr.d_prims; c->d_meta = lr.d_meta;
		for (int i = 0; i < s->n_primitives; i++) devPrimOf[sorted[i]] = i;
		height = lr.height; c->build_ms = lr.build_ms;
		n4nodes = (size_t)4 * lr.n_nodes; n4prims = (size_t)4 * s->n_primitives; nmeta = (size_t)s->n_primitives;
		// the 8-wide tree for the shadow rays, collapsed from the binary tree on the device as well (jp_lbvh.h)
		bool want_wide = s->n_primitives > 64;
		if (const char* ev = getenv("JETPBRT_DEVICE_WIDE")) want_wide = atoi(ev) != 0;
		if (want_wide)
		{
			WideResult wr;
			e = lbvh_build_wide(c->stream, (const float4*)c->d_nodes, s->n_primitives, wr);
			if (e != hipSuccess) return fail(JP_ERR_DEVICE, std::string("jp_upload_scene: device wide-tree build failed: ") + hipGetErrorString(e));
			if (wr.d_wide) { c->d_wide = wr.d_wide; dev_wide = true; dev_n_wide = wr.n_wide; wide_height = wr.height; use_wide = true; c->build_ms += wr.build_ms; }
		}
		// [round 3] ... and the 4-wide tree of Walker<4> for the closest-hit (and shadow) rays, as the host path has it
		bool want_q4 = s->n_primitives > 1024;
		if (const char* ev = getenv("JETPBRT_Q4")) want_q4 = want_q4 && atoi(ev) != 0;
		if (want_q4)
		{
			WideResult qr;
			e = lbvh_build_q4(c->stream, (const float4*)c->d_nodes, s->n_primitives, qr);
			if (e != hipSuccess) return fail(JP_ERR_DEVICE, std::string("jp_upload_scene: device 4-wide tree build failed: ") + hipGetErrorString(e));
			if (qr.d_wide) { c->d_q4 = qr.d_wide; dev_q4 = true; dev_n_q4 = qr.n_wide; c->build_ms += qr.build_ms; use_q4 = true; q4_height = qr.height; }
		}
	}
	c->bvh_height = height; c->bvh_nodes = ref_sem ? s->n_bvh_nodes : (int)(n4nodes / 4);

	// tiny scenes: the flat leaf list of flat_boxes (leaf boxes padded like the node boxes, each with the bit set of its primitives)
	std::vector<float4> flat;
	if (!device_build && !ref_sem && s->n_primitives <= 64)
	{
		int nleaves = 0; for (int n = 0; n < s->n_bvh_nodes; n++) if (seen[n] && s->bvh_left[n] < 0) nleaves++;
		if (nleaves <= 32)
		{
			for (int n = 0; n < s->n_bvh_nodes; n++)
			{
				if (!seen[n] || s->bvh_left[n] >= 0) continue;
				float bb[6]; pad_box(n, bb);
				int first = -s->bvh_left[n] - 1, cnt = s->bvh_right[n];
				unsigned long long bits = 0;
				for (int k = 0; k < cnt; k++) bits |= 1ull << devPrimOf[s->bvh_prim_index[first + k]];
				const uint32_t lo = (uint32_t)bits, hi = (uint32_t)(bits >> 32); float flo, fhi; std::memcpy(&flo, &lo, 4); std::memcpy(&fhi, &hi, 4);
				flat.push_back(make_float4(bb[0], bb[1], bb[2], flo)); flat.push_back(make_float4(bb[3], bb[4], bb[5], fhi));
			}
		}
	}

	// other scenes: the "cut" -- boxes of the largest subtrees below the root (each >= 64 primitives, at most 16), found by opening
	// the heaviest subtree again and again.  How many of them a ray's segment enters is the expected-work key of the sorted
	// traversal kernels (k_extend_sort / k_shadow_sort); the boxes use the flat_boxes layout, one bit each.
	std::vector<float4> cut;
	if (!device_build && flat.empty() && s->bvh_left[0] >= 0)
	{
		std::vector<int> nprim(s->n_bvh_nodes, 0);
		{   // primitives per subtree, children before parents (explicit post-order: the tree may be 30+ levels deep, not 1e5)
			std::vector<std::pair<int, int>> st; st.push_back({ 0, 0 });
			while (!st.empty())
			{
				auto [nd, phase] = st.back(); st.pop_back();
				if (s->bvh_left[nd] < 0) { nprim[nd] = s->bvh_right[nd]; continue; }
				if (phase == 0) { st.push_back({ nd, 1 }); st.push_back({ s->bvh_left[nd], 0 }); st.push_back({ s->bvh_right[nd], 0 }); }
				else nprim[nd] = nprim[s->bvh_left[nd]] + nprim[s->bvh_right[nd]];
			}
		}
		std::vector<int> open; open.push_back(0);
		for (;;)
		{
			int best = -1;
			for (size_t k = 0; k < open.size(); k++) if (s->bvh_left[open[k]] >= 0 && nprim[open[k]] >= 128 && (best < 0 || nprim[open[k]] > nprim[open[best]])) best = (int)k;
			if (best < 0 || open.size() >= 16) break;
			const int nd = open[best]; open.erase(open.begin() + best);
			open.push_back(s->bvh_left[nd]); open.push_back(s->bvh_right[nd]);
		}
		int bit = 0;
		for (int nd : open)
		{
			if (nprim[nd] < 64) continue;
			float bb[6]; pad_box(nd, bb);
			const uint32_t lo = 1u << bit++; float flo; std::memcpy(&flo, &lo, 4);
			cut.push_back(make_float4(bb[0], bb[1], bb[2], flo)); cut.push_back(make_float4(bb[3], bb[4], bb[5], 0.f));
		}
	}

	// materials: the 16-float rows as 4 x float4
	std::vector<float4> mats(4 * std::max(1, s->n_materials)); std::vector<int> mtype(std::max(1, s->n_materials), 0);
	for (int i = 0; i < s->n_materials; i++) { std::memcpy(&mats[4 * i], s->mat_params + (size_t)i * JP_MAT_PARAM_STRIDE, 16 * sizeof(float)); mtype[i] = s->mat_type[i]; }
	// lights: (radiance, type) (device prim, 1/Area(), -, -); areas with the reference's expressions (shape.h:351, 457, 546)
	std::vector<float4> lights(2 * std::max(1, s->n_lights)); int planes = 0, nenv = 0; float envsum[3] = { 0, 0, 0 };
	for (int i = 0; i < s->n_lights; i++)
	{
		int ty = s->light_type[i]; float tf; std::memcpy(&tf, &ty, 4);
		const float* rad = s->light_radiance + 3 * i;
		lights[2 * i] = make_float4(rad[0], rad[1], rad[2], tf);
		bool black = rad[0] == 0.f && rad[1] == 0.f && rad[2] == 0.f;
		if (!black) planes++;
		float inv_area = 0.f; int dp = -1;
		if (ty == JP_LIGHT_AREA)
		{
			int p = s->light_prim[i]; dp = devPrimOf[p];
			int t = s->prim_shape_type[p], k = s->prim_shape_index[p]; float area;
			if (t == JP_SHAPE_TRIANGLE) area = 0.5f * hlen(hcross(hsub(hld(s->tri_p1 + 3 * k), hld(s->tri_p0 + 3 * k)), hsub(hld(s->tri_p2 + 3 * k), hld(s->tri_p0 + 3 * k))));
			else if (t == JP_SHAPE_RECTANGLE) area = hlen(hcross(hsub(hld(s->rect_p0 + 3 * k), hld(s->rect_p1 + 3 * k)), hsub(hld(s->rect_p2 + 3 * k), hld(s->rect_p1 + 3 * k))));
			else if (t == JP_SHAPE_DISK) { const float kPi = (float)3.14159265358979323846; area = kPi * s->disk_radius[k] * s->disk_radius[k]; }   // shape.h:253
			else { const float kPi = (float)3.14159265358979323846; float r2 = s->sph_radius[k] * s->sph_radius[k]; area = 4 * kPi * r2; }
			inv_area = 1 / area;
		}
		else if (ty == JP_LIGHT_ENVIRONMENT) { nenv++; envsum[0] += rad[0]; envsum[1] += rad[1]; envsum[2] += rad[2]; }
		float df; std::memcpy(&df, &dp, 4);
		lights[2 * i + 1] = make_float4(df, inv_area, 0, 0);
		if (ty == JP_LIGHT_POINT || ty == JP_LIGHT_DIRECTION) lights[2 * i + 1] = make_float4(s->light_vec[3 * i], s->light_vec[3 * i + 1], s->light_vec[3 * i + 2], 0);
	}
	// meta.z must index lights (already does); fix nothing else.

	auto up = [&](void** dst, const void* src, size_t bytes) -> hipError_t {
		hipError_t e = hipMalloc(dst, std::max<size_t>(bytes, 16)); if (e != hipSuccess) return e;
		return hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
	};
	if (!device_build)
	{
		HIP_TRY(up(&c->d_nodes, nodes.data(), nodes.size() * sizeof(float4)));
		HIP_TRY(up(&c->d_prims, prims.data(), prims.size() * sizeof(float4)));
		HIP_TRY(up(&c->d_meta, meta.data(), meta.size() * sizeof(int4)));
	}
	HIP_TRY(up(&c->d_mats, mats.data(), mats.size() * sizeof(float4)));
	HIP_TRY(up(&c->d_mat_type, mtype.data(), mtype.size() * sizeof(int)));
	HIP_TRY(up(&c->d_lights, lights.data(), lights.size() * sizeof(float4)));
	{   // k_shade's LDS tables as one array (SceneView::shade_tab); the primitive part only when the host has the records
		std::vector<float4> tabv;
		tabv.insert(tabv.end(), lights.begin(), lights.begin() + 2 * (size_t)s->n_lights);          // exactly the counts the kernel indexes with
		tabv.insert(tabv.end(), mats.begin(), mats.begin() + 4 * (size_t)s->n_materials);
		const size_t at = tabv.size(); tabv.resize(at + ((size_t)s->n_materials + 3) / 4, make_float4(0, 0, 0, 0));
		if (s->n_materials > 0) std::memcpy(&tabv[at], mtype.data(), (size_t)s->n_materials * sizeof(int));
		if (!device_build)
		{
			tabv.insert(tabv.end(), prims.begin(), prims.end());
			const size_t am = tabv.size(); tabv.resize(am + meta.size());
			std::memcpy(&tabv[am], meta.data(), meta.size() * sizeof(int4));
			// FFrame(normal) (geometry.h:345-349, 371-376) of every flat primitive's stored normal, operation by operation as
			// frame_from_z does it on the device (this file is compiled with -ffp-contract=off for the host too)
			auto hnorm = [](HV3 a) { const float l = hlen(a); HV3 r = { a.x / l, a.y / l, a.z / l }; return r; };
			for (size_t pi = 0; pi < meta.size(); pi++)
			{
				const float4 g3 = prims[4 * pi + 3], g1 = prims[4 * pi + 1];
				int type; std::memcpy(&type, &g3.w, 4);
				const HV3 nn = type == JP_SHAPE_DISK ? HV3{ g1.x, g1.y, g1.z } : HV3{ g3.x, g3.y, g3.z };
				const HV3 n = hnorm(nn);
				const HV3 tmp = std::fabs(n.x) > 0.99f ? HV3{ 0, 1, 0 } : HV3{ 1, 0, 0 };
				const HV3 t = hnorm(hcross(n, tmp)), sv = hnorm(hcross(t, n));
				tabv.push_back(make_float4(n.x, n.y, n.z, 0)); tabv.push_back(make_float4(sv.x, sv.y, sv.z, 0)); tabv.push_back(make_float4(t.x, t.y, t.z, 0));
			}
		}
		HIP_TRY(up(&c->d_shade_tab, tabv.data(), tabv.size() * sizeof(float4)));
	}
	if (use_wide && !dev_wide) HIP_TRY(up(&c->d_wide, wide.data(), wide.size() * sizeof(uint32_t)));
	if ((use_q4 && !dev_q4) || use_cert) HIP_TRY(up(&c->d_q4, q4.data(), q4.size() * sizeof(uint32_t)));
	if (use_cert) HIP_TRY(up(&c->d_refbox, refbox.data(), refbox.size() * sizeof(float4)));
	if (!flat.empty()) HIP_TRY(up(&c->d_flat, flat.data(), flat.size() * sizeof(float4)));
	if (!cut.empty()) HIP_TRY(up(&c->d_cut, cut.data(), cut.size() * sizeof(float4)));

	SceneView& v = c->sv;
	v.nodes = (const float4*)c->d_nodes; v.n_nodes = (int)(n4nodes / 4);
	v.prims = (const float4*)c->d_prims; v.meta = (const int4*)c->d_meta; v.n_prims = (int)nmeta;
	v.mats = (const float4*)c->d_mats; v.mat_type = (const int*)c->d_mat_type; v.n_mats = s->n_materials;
	v.lights = (const float4*)c->d_lights; v.n_lights = s->n_lights; v.shade_tab = (const float4*)c->d_shade_tab;
	v.env_sum = make_float3(envsum[0], envsum[1], envsum[2]); v.n_env = nenv;
	v.world_radius = s->world_radius; v.cam = s->camera;
	v.flat = (const float4*)c->d_flat; v.n_flat = (int)(flat.size() / 2);
	v.cut = (const float4*)c->d_cut; v.n_cut = (int)(cut.size() / 2);
	v.wide = (const uint4*)c->d_wide; v.n_wide = dev_wide ? dev_n_wide : (int)(wide.size() / 20);
	v.q4 = (const uint4*)c->d_q4; v.n_q4 = dev_q4 ? dev_n_q4 : (int)(q4.size() / 16);
	v.refbox = (const float4*)c->d_refbox; v.cert_pad = cert_pad; v.cert_pad_eye = cert_pad_eye; c->cert = use_cert;
	c->dual = false; if (const char* e = getenv("JETPBRT_DUAL")) c->dual = atoi(e) != 0;
	c->trav_lds_pad = 0; if (const char* e = getenv("JETPBRT_TRAV_LDS_PAD")) { const long v = atol(e); if (v > 0 && v <= 48 * 1024) c->trav_lds_pad = (size_t)v & ~(size_t)15; }
	c->use_q4 = use_q4; c->q4_shadow = use_q4;                       // shadow rays too (measured against the 8-wide tree: k_shadow 53.8 -> 52.7 ms per 512 spp, frame +4 %)
	if (const char* e = getenv("JETPBRT_Q4_SHADOW")) c->q4_shadow = use_q4 && atoi(e) != 0;
	c->stack_depth = std::max(2, height + 2);
	if (use_q4 || use_cert) c->stack_depth = std::max(c->stack_depth, 3 * q4_height + 2);      // a 4-wide node pushes up to three children
	size_t scene_bytes = (n4nodes + n4prims) / 4 * 5 * sizeof(float4);   // 80-byte LDS record stride
	size_t prim_bytes = n4prims / 4 * 5 * sizeof(float4);
	size_t stack_bytes = (size_t)c->stack_depth * JP_BLOCK * sizeof(int);
	c->scene_in_lds = !device_build && scene_bytes + stack_bytes <= 40 * 1024;   // device-built trees are indexed sparsely (Karras numbering): global memory only
	c->trav_mode = use_wide ? 3 : ((!flat.empty() && prim_bytes <= 40 * 1024) ? 2 : (c->scene_in_lds ? 1 : 0));
	if (!use_wide) if (const char* e = getenv("JETPBRT_TRAVERSAL")) { int m = atoi(e); if (m == 0 || (m == 1 && c->scene_in_lds)) c->trav_mode = m; }   // experiments: force a lower mode
	if (use_wide) c->scene_in_lds = false;
	if (ref_sem) c->trav_mode = 5;
	// large scenes: closest-hit rays walk the binary tree (exact near-to-far order, early out), any-hit shadow rays the
	// 8-wide quantised tree (fewest node fetches; order irrelevant).  Measured on the 280k-triangle scene:
	// k_extend 10.3 ms binary vs 13.8 ms wide, k_shadow 10.6 ms binary vs 8.6 ms wide.
	c->lds_bytes_shadow = c->trav_mode == 3 ? (size_t)2 * (wide_height + 2) * JP_BLOCK * sizeof(int) : 0;
	c->lds_bytes = c->trav_mode == 2 ? prim_bytes : (c->trav_mode == 1 ? stack_bytes + scene_bytes : stack_bytes);
	if (c->trav_mode != 3) c->lds_bytes_shadow = c->lds_bytes;
	{
		size_t tab = ((size_t)2 * s->n_lights + (size_t)4 * s->n_materials) * sizeof(float4) + (size_t)s->n_materials * sizeof(int) + 16;
		c->tables_in_lds = tab <= 16 * 1024;
		// k_shade's static LDS (tile index, keys, counters of the material sort) + tables + staging must stay within 64 KB a workgroup;
		// beyond 24 KB of tables the kernel's three workgroups per CU would not fit the CU's LDS either
		const size_t shade_static = (size_t)JP_SHADE_TILE * 3 + (size_t)JP_SHADE_CLASSES * (JP_SHADE_TILE / JP_BLOCK) * (JP_BLOCK / 64) * 4 + 128;
		const size_t prim_part = n4prims * sizeof(float4) + nmeta * sizeof(int4) + 3 * nmeta * sizeof(float4);     // records, meta, shading frames
		c->shade_prims_in_lds = c->tables_in_lds && c->scene_in_lds && tab + prim_part <= 24 * 1024;
		c->shade_lds_bytes = c->tables_in_lds ? tab + (c->shade_prims_in_lds ? prim_part : 0) : 0;
		const size_t stage_bytes = 16 + (size_t)std::max(1, planes) * 2 * JP_BLOCK * sizeof(float4);
		c->stage_nee = c->tables_in_lds && std::max(1, planes) <= 4 && shade_static + c->shade_lds_bytes + stage_bytes <= 64 * 1024;
		if (c->stage_nee) c->shade_lds_bytes += stage_bytes;
	}
	c->n_planes = std::max(1, planes);
	{   // material sort in k_shade: pays when the primitives carry more than one material kind (JETPBRT_SHADE_SORT = 0 / 1 forces it)
		bool kinds[8] = { false, false, false, false, false, false, false, false }; int nk = 0;
		for (int i = 0; i < s->n_primitives; i++) { const int m = s->prim_material[i]; const int k = m < 0 ? 7 : s->mat_type[m]; if (!kinds[k]) { kinds[k] = true; nk++; } }
		if (const char* e = getenv("JETPBRT_STACK_LDS")) { const int v = atoi(e); if (v >= 2) c->stack_lds_words = v & ~1; }   // even: the wide tree's entries are word pairs
		// lane refill in the traversal kernels (k_extend_persist / k_shadow_persist): on by default for scenes walked through global
		// memory (measured on the 280k-triangle scene: k_extend 39.1 -> 28.4 ms, k_shadow 28.8 -> 18.9 ms per 128 spp; reference-tree
		// mode 154 -> 227 Msamples/s); the LDS-resident Cornell box loses with it (reference-tree mode 1109 -> 965), so small scenes keep
		// the one-ray-per-lane kernels.  JETPBRT_PERSIST = 0 (off) or the refill threshold (8 / 16 / 32 idle lanes).
		c->persist = ((c->trav_mode == 0 || c->trav_mode == 3 || c->trav_mode == 5) && s->n_primitives > 1024) ? 16 : 0;
		if (const char* e = getenv("JETPBRT_PERSIST")) c->persist = atoi(e);
		// each iteration the lanes of a wave vote on the kind of step it runs (node / leaf); measured on the 280k-triangle scene: k_extend
		// 28.3 -> 21.9 ms, k_shadow 18.9 -> 16.5 ms per 128 spp.  The reference-tree walk (one node per step, leaf objects as their own
		// steps) is faster without it: 310 vs 286 Msamples/s.
		c->vote = c->trav_mode != 5 || c->cert; if (const char* e = getenv("JETPBRT_VOTE")) c->vote = atoi(e) != 0;
		c->ray_sort = false;                                        // opt-in: JETPBRT_RAY_SORT=1 (tiny scenes: by primitive-test count; others: by cut boxes entered)
		if (const char* e = getenv("JETPBRT_RAY_SORT")) c->ray_sort = atoi(e) != 0 && (c->trav_mode == 2 || ((c->trav_mode == 0 || c->trav_mode == 3) && !cut.empty()));
		c->shade_sort = nk > 1;
		c->class_mask = 1; for (int k = 0; k < 5; k++) if (kinds[k]) c->class_mask |= 2 << k;
		if (const char* e = getenv("JETPBRT_SHADE_SORT")) c->shade_sort = atoi(e) != 0;
	}
	c->has_null_material = hasNull;
	c->have_scene = true;
	return JP_OK;
}

// ---- render ---------------------------------------------------------------------------------------------------------------
namespace
{
enum { CLS_EXTEND = 0, CLS_SHADE = 1, CLS_SHADOW = 2, CLS_OTHER = 3, CLS_PATH = 4 };

int ensure_queues(JpContext* c, unsigned int cap, int planes, unsigned int nblocks)
{
	if (c->cap >= cap && c->planes_alloc >= planes && c->blk_alloc >= nblocks) return JP_OK;
	cap = std::max(cap, c->cap); planes = std::max(planes, c->planes_alloc); nblocks = std::max(nblocks, c->blk_alloc);
	free_queues(c); c->blk_alloc = 0;
	auto alloc = [&](void** p, size_t bytes) -> bool { if (hipMalloc(p, bytes) != hipSuccess) return false; c->qbufs.push_back(*p); return true; };
	Queues& q = c->q; bool ok = true;
	for (int b = 0; b < 2 && ok; b++) ok = alloc((void**)&q.ray_o[b], (size_t)cap * 16) && alloc((void**)&q.ray_d[b], (size_t)cap * 16) && alloc((void**)&q.beta[b], (size_t)cap * 16)
	                                       && alloc((void**)&q.blk_q[b], (size_t)nblocks * 4);
	ok = ok && alloc((void**)&q.hit, (size_t)cap * 8) && alloc((void**)&q.lacc, (size_t)cap * 16) && alloc((void**)&q.sh_o, (size_t)cap * 16)
	     && alloc((void**)&q.sh_d, (size_t)cap * 16 * planes) && alloc((void**)&q.sh_c, (size_t)cap * 16 * planes) && alloc((void**)&q.blk_sh, (size_t)nblocks * 4);
	if (!ok) { free_queues(c); return fail(JP_ERR_DEVICE, "jp_render: out of device memory for the path queues"); }
	c->cap = cap; c->planes_alloc = planes; c->blk_alloc = nblocks;
	return JP_OK;
}

struct Stamper
{
	JpContext* c; int cls; size_t a; hipStream_t st;
	Stamper(JpContext* c, int cls, hipStream_t st_ = nullptr) : c(c), cls(cls), a(0), st(st_ ? st_ : c->stream)
	{
		if (!c->profiling) return;
		if (c->evused + 2 > c->evpool.size()) { size_t old = c->evpool.size(); c->evpool.resize(old + 64); for (size_t i = old; i < c->evpool.size(); i++) hipEventCreate(&c->evpool[i]); }
		a = c->evused; c->evused += 2;
		hipEventRecord(c->evpool[a], st);
	}
	~Stamper() { if (!c->profiling) return; hipEventRecord(c->evpool[a + 1], st); JpContext::Stamp s = { cls, a, a + 1 }; c->stamps.push_back(s); }
};

int render_one(JpContext* c, const JpRenderParams* rp, float* film_dev, bool sync, int lane_index = 0, int lane_count = 1, int lane_group = 4, bool ev0_recorded = false)
{
	if (!c || !rp || !film_dev) return fail(JP_ERR_INVALID_ARGUMENT, "jp_render: null argument");
	if (!c->have_scene) return fail(JP_ERR_NO_SCENE, "jp_render: no scene uploaded");
	if (rp->width <= 0 || rp->height <= 0 || rp->spp <= 0 || rp->max_depth < 0 || rp->max_depth > 200) return fail(JP_ERR_INVALID_ARGUMENT, "jp_render: bad width/height/spp/max_depth");
	if (rp->integrator < JP_INTEGRATOR_PATH || rp->integrator > JP_INTEGRATOR_DEBUG_NORMAL) return fail(JP_ERR_INVALID_ARGUMENT, "jp_render: unknown integrator");
	if (rp->integrator == JP_INTEGRATOR_WHITTED && rp->max_depth > JP_WHITTED_MAX_DEPTH) return fail(JP_ERR_UNSUPPORTED, "jp_render: the Whitted integrator supports max_depth <= 16");
	if (rp->sampler_mode != JP_SAMPLER_COUNTER && rp->sampler_mode != JP_SAMPLER_DEBUG) return fail(JP_ERR_UNSUPPORTED, "jp_render: the device path implements the counter sampler only (the sequential mt19937_64 stream is not reproducible in parallel)");
	const int band = rp->band_rows > 0 ? rp->band_rows : 20;
	const int scount = rp->shard_count > 1 ? rp->shard_count : 1;
	const int sidx = scount > 1 ? rp->shard_index : 0;
	if (sidx < 0 || sidx >= scount) return fail(JP_ERR_INVALID_ARGUMENT, "jp_render: shard_index out of range");
	HIP_TRY(hipSetDevice(c->device));

	const int nbands = (rp->height + band - 1) / band;
	int local_rows = 0;
	for (int b = sidx; b < nbands; b += scount) local_rows += std::min(band, rp->height - b * band);
	if (lane_count > 1)
	{   // this lane's share of the shard's rows: groups of lane_group rows dealt round-robin (only the shard's last group can be short)
		int mine = 0;
		for (int g0 = lane_index * lane_group; g0 < local_rows; g0 += lane_count * lane_group) mine += std::min(lane_group, local_rows - g0);
		local_rows = mine;
	}
	const long long npix = (long long)local_rows * rp->width;

	if (!ev0_recorded) HIP_TRY(hipEventRecord(c->ev0, c->stream));          // (with several lanes render_impl records it before the first lane is enqueued)
	HIP_TRY(hipMemsetAsync(film_dev, 0, sizeof(float) * 3 * (size_t)rp->width * rp->height, c->stream));
	HIP_TRY(hipMemsetAsync(c->d_cnt, 0, sizeof(DevCounters), c->stream));
	c->evused = 0; c->stamps.clear();
	unsigned long long samples = 0;
	if (npix > 0)
	{
		if (npix > (1 << 24)) return fail(JP_ERR_UNSUPPORTED, "jp_render: more than 2^24 pixels per shard");
		// a shadow entry's header packs (slot, ray count) in 32 bits: 27 + 5 as a rule; batches of up to 2^26 slots (regions of <= 8192
		// slots: up to 8192 workgroups a launch, whose tail -- the last workgroups finishing on an emptying GPU -- weighs a quarter of
		// what it does with 2^24)
		const int slot_bits = c->n_planes <= 31 ? 27 : 24;
		const unsigned int PMAX = slot_bits == 27 ? (1u << 26) : (1u << 24);
		// memory budget for the queues: ~ (136 + 32 * planes) bytes per slot.  ONE budget -- half of what is free, at most 24 GB per lane --
		// shared by the lanes that render concurrently (each lane sizes its own queue set from its share), and when the allocation still
		// fails (another process took the memory in between) the batch is halved and tried again before the call gives up
		size_t freeB = 0, totalB = 0; hipMemGetInfo(&freeB, &totalB);
		const size_t per = 136 + 32 * (size_t)c->n_planes;
		size_t budget = std::min<size_t>((size_t)24 << 30, (freeB / (size_t)std::max(1, lane_count) + (c->cap ? (size_t)c->cap * (136 + 32 * (size_t)c->planes_alloc) : 0)) / 2);
		if (const char* e = getenv("JETPBRT_MAX_SLOTS")) { long long v = atoll(e); if (v > 0) budget = std::min<size_t>(budget, (size_t)v * per); }
		int sbatch = 1; unsigned int P = 0, G = 1, R = JP_BLOCK, cap = 0;
		for (int attempt = 0;; attempt++)
		{
			const unsigned int pcap = (unsigned int)std::min<size_t>(PMAX, std::max<size_t>((size_t)npix, budget / per));
			sbatch = (int)std::max<long long>(1, std::min<long long>(rp->spp, pcap / npix));
			{   // equal batches: ceil(spp / sbatch) batches of (nearly) the same size instead of full ones and a remainder (1024 spp in batches of
				// 192 would end with a 64-spp batch whose launches fill a third of the GPU)
				const int nb = (rp->spp + sbatch - 1) / sbatch;
				sbatch = (rp->spp + nb - 1) / nb;
			}
			if ((long long)sbatch * npix > (long long)PMAX) return fail(JP_ERR_UNSUPPORTED, "jp_render: shard too large for one batch");
			P = (unsigned int)((long long)sbatch * npix);
			const unsigned int nchunks = (P + JP_BLOCK - 1) / JP_BLOCK;
			G = std::max(1u, std::min(nchunks, (unsigned int)(c->n_cus * c->blocks_per_cu)));   // one region per workgroup
			G = std::max(G, (nchunks + JP_SHADE_TILE / JP_BLOCK - 1) / (JP_SHADE_TILE / JP_BLOCK));   // R <= JP_SHADE_TILE: k_shade partitions a whole region in LDS and counts its fills in 16 bits
			R = ((nchunks + G - 1) / G) * JP_BLOCK;
			cap = G * R;
			const int st = ensure_queues(c, cap, c->n_planes, G);
			if (st == JP_OK) break;
			if (sbatch <= 1 || attempt >= 6) return st;                 // one sample per pixel does not fit either: out of device memory
			budget = (size_t)sbatch / 2 * (size_t)npix * per;              // half the batch
		}
		c->q.cap = cap; c->q.R = R;
		{   // spill area of the walkers' stacks: the words a thread may need beyond the ones kept in LDS
			const int deep = std::max(c->stack_depth, c->trav_mode == 3 ? (int)(c->lds_bytes_shadow / (JP_BLOCK * sizeof(int))) : 0);
			const size_t need = c->persist && deep >= c->stack_lds_words ? (size_t)(deep - c->stack_lds_words + 1) * G * JP_BLOCK : 1;   // (+1: Walker<4> keeps one LDS word as a dump slot)
			if (c->spill_words < need) { if (c->d_spill) hipFree(c->d_spill); c->d_spill = nullptr; HIP_TRY(hipMalloc((void**)&c->d_spill, need * sizeof(int))); c->spill_words = need; }
			if (c->dual && c->persist)
			{
				if (c->spill2_words < need) { if (c->d_spill2) hipFree(c->d_spill2); c->d_spill2 = nullptr; HIP_TRY(hipMalloc((void**)&c->d_spill2, need * sizeof(int))); c->spill2_words = need; }
				if (!c->stream2) { HIP_TRY(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking)); HIP_TRY(hipEventCreateWithFlags(&c->ev_shade, hipEventDisableTiming)); HIP_TRY(hipEventCreateWithFlags(&c->ev_shadow, hipEventDisableTiming)); }
			}
		}
		if (c->pix_acc_n < (size_t)npix) { if (c->d_pix_acc) hipFree(c->d_pix_acc); c->d_pix_acc = nullptr; HIP_TRY(hipMalloc((void**)&c->d_pix_acc, (size_t)npix * 16)); c->pix_acc_n = (size_t)npix; }

		RenderConst rc; rc.width = rp->width; rc.height = rp->height; rc.spp = rp->spp; rc.max_depth = rp->max_depth; rc.seed = rp->seed;
		rc.band_rows = band; rc.shard_index = sidx; rc.shard_count = scount; rc.npix = (int)npix; rc.local_rows = local_rows; rc.n_planes = c->n_planes;
		rc.lane_index = lane_index; rc.lane_count = lane_count; rc.lane_rows = lane_group; rc.class_mask = c->class_mask; rc.sampler_debug = rp->sampler_mode == JP_SAMPLER_DEBUG ? 1 : 0;
		// measured: +6 % on the 280k-triangle scene (cache reuse), -8 % on the LDS-resident Cornell box (coherent waves finish
		// together or not at all, which unbalances the workgroups) -> tiles only when traversal goes through global memory
		rc.slot_bits = slot_bits;
		rc.tiled = (c->trav_mode == 0 && rp->width % 16 == 0 && local_rows % 4 == 0 && !getenv("JETPBRT_NO_TILES")) ? 1 : 0;
		// compact regions (k_raygen): scenes walked through global memory -- one lane on the 280k-triangle scene: k_extend 64.1 -> 55.8 ms,
		// k_shadow 47.1 -> 40.5 ms per 512 spp (the workgroups in flight share an image area, hence tree nodes: L2), three lanes +1.2 %;
		// films bit-identical.  JETPBRT_COMPACT_REGIONS=0 / 1 forces it.
		rc.compact = (c->trav_mode != 2 && c->trav_mode != 1 && npix % JP_BLOCK == 0) ? 1 : 0;
		if (const char* e = getenv("JETPBRT_COMPACT_REGIONS")) rc.compact = (atoi(e) != 0 && npix % JP_BLOCK == 0) ? 1 : 0;
		const int grid = (int)G;
		const size_t lds = c->lds_bytes;
		for (int s0 = 0; s0 < rp->spp; s0 += sbatch)
		{
			rc.s0 = s0; rc.sbatch = std::min(sbatch, rp->spp - s0);
			{ Stamper t(c, CLS_OTHER); hipLaunchKernelGGL(k_raygen, dim3(grid), dim3(JP_BLOCK), 0, c->stream, c->sv, c->q, rc, c->d_cnt); }
			if (rp->integrator != JP_INTEGRATOR_PATH)
			{   // the other two integrators: one megakernel launch per batch (k_other), then the same per-pixel sum
				Stamper t(c, CLS_OTHER);
				const int ogrid = (int)std::min<unsigned int>((P + JP_BLOCK - 1) / JP_BLOCK, (unsigned int)(c->n_cus * 16));
				const size_t stack_lds = (size_t)c->stack_depth * JP_BLOCK * sizeof(int);
				if (c->trav_mode == 5) hipLaunchKernelGGL(k_other<5>, dim3(ogrid), dim3(JP_BLOCK), stack_lds, c->stream, c->sv, c->q, rc, rp->integrator, c->stack_depth, c->d_cnt);
				else if (c->trav_mode == 2) hipLaunchKernelGGL(k_other<2>, dim3(ogrid), dim3(JP_BLOCK), lds, c->stream, c->sv, c->q, rc, rp->integrator, c->stack_depth, c->d_cnt);
				else if (c->trav_mode == 1) hipLaunchKernelGGL(k_other<1>, dim3(ogrid), dim3(JP_BLOCK), lds, c->stream, c->sv, c->q, rc, rp->integrator, c->stack_depth, c->d_cnt);
				else hipLaunchKernelGGL(k_other<0>, dim3(ogrid), dim3(JP_BLOCK), stack_lds, c->stream, c->sv, c->q, rc, rp->integrator, c->stack_depth, c->d_cnt);
			}
			int cur = 0; bool shadow_pending = false;
			int iters = rp->integrator != JP_INTEGRATOR_PATH ? 0 : rp->max_depth + 1;
			for (int it = 0;; it++)
			{
				if (it >= iters)
				{
					if (rp->integrator != JP_INTEGRATOR_PATH || !c->has_null_material || it > iters + 64) break;
					// null-material primitives re-queue a path without consuming a bounce (integrator.cc:349-353): ask the device
					DevCounters h; HIP_TRY(hipMemcpyAsync(&h, c->d_cnt, sizeof(h), hipMemcpyDeviceToHost, c->stream)); HIP_TRY(hipStreamSynchronize(c->stream));
					if (h.n_queue[cur] == 0) break;
				}
				{
					Stamper t(c, CLS_EXTEND);
					if (c->persist && (c->trav_mode == 0 || c->trav_mode == 3 || c->trav_mode == 5))
					{
						const int ecap = std::min(c->stack_depth, c->stack_lds_words); const size_t elds = (size_t)ecap * JP_BLOCK * sizeof(int) + c->trav_lds_pad;
						#define JP_LAUNCH_EP(M, R) do { if (c->vote) hipLaunchKernelGGL((k_extend_persist<M, R, true>), dim3(grid), dim3(JP_BLOCK), elds, c->stream, c->sv, c->q, cur, ecap, c->d_spill, c->d_cnt); else hipLaunchKernelGGL((k_extend_persist<M, R, false>), dim3(grid), dim3(JP_BLOCK), elds, c->stream, c->sv, c->q, cur, ecap, c->d_spill, c->d_cnt); } while (0)
						if (c->trav_mode == 5 && c->cert) { if (c->persist >= 32) JP_LAUNCH_EP(6, 32); else if (c->persist >= 16) JP_LAUNCH_EP(6, 16); else JP_LAUNCH_EP(6, 8); }
						else if (c->trav_mode == 5) { if (c->persist >= 32) JP_LAUNCH_EP(5, 32); else if (c->persist >= 16) JP_LAUNCH_EP(5, 16); else JP_LAUNCH_EP(5, 8); }
						else if (c->use_q4) { if (c->persist >= 32) JP_LAUNCH_EP(4, 32); else if (c->persist >= 16) JP_LAUNCH_EP(4, 16); else JP_LAUNCH_EP(4, 8); }
						else { if (c->persist >= 32) JP_LAUNCH_EP(0, 32); else if (c->persist >= 16) JP_LAUNCH_EP(0, 16); else JP_LAUNCH_EP(0, 8); }
						#undef JP_LAUNCH_EP
					}
					else if (c->ray_sort)
					{
						if (c->trav_mode == 2) hipLaunchKernelGGL(k_extend_sort<2>, dim3(grid), dim3(JP_BLOCK), lds, c->stream, c->sv, c->q, cur, c->stack_depth, c->d_cnt);
						else hipLaunchKernelGGL(k_extend_sort<0>, dim3(grid), dim3(JP_BLOCK), lds, c->stream, c->sv, c->q, cur, c->stack_depth, c->d_cnt);
					}
					else if (c->trav_mode == 5) hipLaunchKernelGGL(k_extend<5>, dim3(grid), dim3(JP_BLOCK), lds, c->stream, c->sv, c->q, cur, c->stack_depth, c->d_cnt);
					else if (c->trav_mode == 2) hipLaunchKernelGGL(k_extend<2>, dim3(grid), dim3(JP_BLOCK), lds, c->stream, c->sv, c->q, cur, c->stack_depth, c->d_cnt);
					else if (c->trav_mode == 1) hipLaunchKernelGGL(k_extend<1>, dim3(grid), dim3(JP_BLOCK), lds, c->stream, c->sv, c->q, cur, c->stack_depth, c->d_cnt);
					else hipLaunchKernelGGL(k_extend<0>, dim3(grid), dim3(JP_BLOCK), lds, c->stream, c->sv, c->q, cur, c->stack_depth, c->d_cnt);
				}
				if (shadow_pending) { HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_shadow, 0)); shadow_pending = false; }   // (k_shade rewrites the shadow queues and adds to the paths' radiance)
				{
					Stamper t(c, CLS_SHADE);
					const bool st = c->stage_nee;
					#define JP_LAUNCH_SHADE(A, B, C) do { if (c->shade_sort) hipLaunchKernelGGL((k_shade<A, B, C, true>), dim3(grid), dim3(JP_BLOCK), c->shade_lds_bytes, c->stream, c->sv, c->q, rc, cur, c->d_cnt); \
					                                   else hipLaunchKernelGGL((k_shade<A, B, C, false>), dim3(grid), dim3(JP_BLOCK), c->shade_lds_bytes, c->stream, c->sv, c->q, rc, cur, c->d_cnt); } while (0)
					if (c->shade_prims_in_lds) { if (st) JP_LAUNCH_SHADE(true, true, true); else JP_LAUNCH_SHADE(true, true, false); }
					else if (c->tables_in_lds) { if (st) JP_LAUNCH_SHADE(true, false, true); else JP_LAUNCH_SHADE(true, false, false); }
					else JP_LAUNCH_SHADE(false, false, false);
					#undef JP_LAUNCH_SHADE
				}
				HIP_TRY(hipGetLastError());                               // a failed launch (k_extend / k_shade) is reported where it happens, not at the end of the frame
				if (it < rp->max_depth || c->has_null_material)                 // at bounce == maxDepth Li() breaks before the NEE (integrator.cc:340-343)
				{
					const bool two = c->dual && c->persist && c->stream2 && !c->has_null_material;
					hipStream_t sstream = two ? c->stream2 : c->stream; int* sspill = two ? c->d_spill2 : c->d_spill;
					if (two) { HIP_TRY(hipEventRecord(c->ev_shade, c->stream)); HIP_TRY(hipStreamWaitEvent(c->stream2, c->ev_shade, 0)); }
					Stamper t(c, CLS_SHADOW, sstream);
					const size_t slds = c->q4_shadow ? (size_t)c->stack_depth * JP_BLOCK * sizeof(int) : (c->trav_mode == 3 ? c->lds_bytes_shadow : lds);
					const int scap = std::min((int)(slds / (JP_BLOCK * sizeof(int))), c->stack_lds_words);     // stack words per thread kept in LDS
					const size_t plds = (size_t)scap * JP_BLOCK * sizeof(int) + (((size_t)c->q.R * c->n_planes + 31) / 32) * 4 * (c->cert ? 2 : 1) + c->trav_lds_pad;   // (certified walk: a second bitmap, the rays without a certificate)
					if (c->persist && (c->trav_mode == 0 || c->trav_mode == 3 || c->trav_mode == 5) && plds <= 64 * 1024)
					{
						#define JP_LAUNCH_SP(M, R) do { if (c->vote) hipLaunchKernelGGL((k_shadow_persist<M, R, true>), dim3(grid), dim3(JP_BLOCK), plds, sstream, c->sv, c->q, rc, scap, sspill, c->d_cnt); else hipLaunchKernelGGL((k_shadow_persist<M, R, false>), dim3(grid), dim3(JP_BLOCK), plds, sstream, c->sv, c->q, rc, scap, sspill, c->d_cnt); } while (0)
						if (c->trav_mode == 5 && c->cert) { if (c->persist >= 32) JP_LAUNCH_SP(6, 32); else if (c->persist >= 16) JP_LAUNCH_SP(6, 16); else JP_LAUNCH_SP(6, 8); }
						else if (c->trav_mode == 5) { if (c->persist >= 32) JP_LAUNCH_SP(5, 32); else if (c->persist >= 16) JP_LAUNCH_SP(5, 16); else JP_LAUNCH_SP(5, 8); }
						else if (c->q4_shadow) { if (c->persist >= 32) JP_LAUNCH_SP(4, 32); else if (c->persist >= 16) JP_LAUNCH_SP(4, 16); else JP_LAUNCH_SP(4, 8); }
						else if (c->trav_mode == 3) { if (c->persist >= 32) JP_LAUNCH_SP(3, 32); else if (c->persist >= 16) JP_LAUNCH_SP(3, 16); else JP_LAUNCH_SP(3, 8); }
						else { if (c->persist >= 32) JP_LAUNCH_SP(0, 32); else if (c->persist >= 16) JP_LAUNCH_SP(0, 16); else JP_LAUNCH_SP(0, 8); }
						#undef JP_LAUNCH_SP
					}
					else if (c->ray_sort)
					{
						if (c->trav_mode == 2) hipLaunchKernelGGL(k_shadow_sort<2>, dim3(grid), dim3(JP_BLOCK), lds, c->stream, c->sv, c->q, rc, c->stack_depth, c->d_cnt);
						else if (c->trav_mode == 3) hipLaunchKernelGGL(k_shadow_sort<3>, dim3(grid), dim3(JP_BLOCK), c->lds_bytes_shadow, c->stream, c->sv, c->q, rc, c->stack_depth, c->d_cnt);
						else hipLaunchKernelGGL(k_shadow_sort<0>, dim3(grid), dim3(JP_BLOCK), lds, c->stream, c->sv, c->q, rc, c->stack_depth, c->d_cnt);
					}
					else if (c->trav_mode == 3) hipLaunchKernelGGL(k_shadow<3>, dim3(grid), dim3(JP_BLOCK), c->lds_bytes_shadow, c->stream, c->sv, c->q, rc, c->stack_depth, c->d_cnt);
					else if (c->trav_mode == 5) hipLaunchKernelGGL(k_shadow<5>, dim3(grid), dim3(JP_BLOCK), lds, c->stream, c->sv, c->q, rc, c->stack_depth, c->d_cnt);
					else if (c->trav_mode == 2) hipLaunchKernelGGL(k_shadow<2>, dim3(grid), dim3(JP_BLOCK), lds, c->stream, c->sv, c->q, rc, c->stack_depth, c->d_cnt);
					else if (c->trav_mode == 1) hipLaunchKernelGGL(k_shadow<1>, dim3(grid), dim3(JP_BLOCK), lds, c->stream, c->sv, c->q, rc, c->stack_depth, c->d_cnt);
					else hipLaunchKernelGGL(k_shadow<0>, dim3(grid), dim3(JP_BLOCK), lds, c->stream, c->sv, c->q, rc, c->stack_depth, c->d_cnt);
					if (two) { HIP_TRY(hipEventRecord(c->ev_shadow, c->stream2)); shadow_pending = true; }
					HIP_TRY(hipGetLastError());
				}
				cur ^= 1;
			}
			if (shadow_pending) { HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_shadow, 0)); shadow_pending = false; }
			{ Stamper t(c, CLS_OTHER); hipLaunchKernelGGL(k_resolve, dim3((unsigned int)std::min<long long>(c->n_cus * 8, (npix + JP_BLOCK - 1) / JP_BLOCK)), dim3(JP_BLOCK), 0, c->stream, c->q, rc, c->d_pix_acc, film_dev, s0 == 0 ? 1 : 0, s0 + rc.sbatch >= rp->spp ? 1 : 0); }
			samples += (unsigned long long)rc.sbatch * (unsigned long long)npix;
		}
		HIP_TRY(hipGetLastError());
	}
	HIP_TRY(hipEventRecord(c->ev1, c->stream));
	c->own_samples = samples;
	if (sync)
	{
		HIP_TRY(hipStreamSynchronize(c->stream));
	}
	return JP_OK;
}

// ---- stream lanes: the shard's bands dealt to L lanes, rendered concurrently on L streams with L queue sets ---------------
__global__ void __launch_bounds__(JP_BLOCK) k_add_film(float* __restrict__ dst, const float* __restrict__ src, size_t n)
{
	// the lanes' films are disjoint (zero outside a lane's bands), so the sum is the union, bit for bit
	for (size_t i = (size_t)blockIdx.x * JP_BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * JP_BLOCK) dst[i] += src[i];
}

int make_lanes(JpContext* c, int extra)
{
	if (!c->ev_added && hipEventCreateWithFlags(&c->ev_added, hipEventDisableTiming) != hipSuccess) return fail(JP_ERR_DEVICE, "jp_render: event allocation failed");
	while ((int)c->lanes.size() < extra)
	{
		JpContext* l = new JpContext;
		l->device = c->device; l->is_lane = true; l->n_cus = c->n_cus; l->blocks_per_cu = c->blocks_per_cu;
		std::memset(&l->counters, 0, sizeof(l->counters)); std::memset(&l->q, 0, sizeof(l->q));
		if (hipStreamCreateWithFlags(&l->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&l->ev0) != hipSuccess || hipEventCreate(&l->ev1) != hipSuccess
		    || hipMalloc((void**)&l->d_cnt, sizeof(DevCounters)) != hipSuccess)
		{ jp_destroy_context(l); return fail(JP_ERR_DEVICE, "jp_render: stream/event allocation for an extra lane failed"); }
		c->lanes.push_back(l);
	}
	return JP_OK;
}

// a lane walks the same device tables as its parent (it owns none of them)
void sync_lane_scene(JpContext* c, JpContext* l)
{
	l->have_scene = c->have_scene; l->sv = c->sv; l->stack_depth = c->stack_depth; l->scene_in_lds = c->scene_in_lds; l->shade_prims_in_lds = c->shade_prims_in_lds;
	l->lds_bytes = c->lds_bytes; l->lds_bytes_shadow = c->lds_bytes_shadow; l->trav_mode = c->trav_mode; l->n_planes = c->n_planes;
	l->stack_lds_words = c->stack_lds_words;
	l->use_q4 = c->use_q4; l->q4_shadow = c->q4_shadow; l->trav_lds_pad = c->trav_lds_pad; l->dual = c->dual; l->cert = c->cert;
	l->ray_sort = c->ray_sort; l->persist = c->persist; l->vote = c->vote; l->shade_sort = c->shade_sort; l->class_mask = c->class_mask;
	l->has_null_material = c->has_null_material; l->tables_in_lds = c->tables_in_lds; l->stage_nee = c->stage_nee; l->shade_lds_bytes = c->shade_lds_bytes;
	l->profiling = c->profiling;
}


// ---- fused schedule: one k_path launch per batch (jp_path.h) --------------------------------------------------------------
// OPT-IN (JETPBRT_FUSED=1; FScene / CLI: --fused).  Measured in round 3 (profiles/r03a_fused_ab.txt): films bit-identical to the
// per-bounce launches, queue memory 1.2 GB instead of 13-40 GB -- and 20 % (Cornell) to 57 % (280k-triangle scene) SLOWER than three
// stream lanes: k_path inherits k_shade's 168 registers, so the traversal phases run at 3 waves per SIMD instead of 8, and a region that
// fits LDS-resident hit records and radiance (1024 paths) gives every phase of a late bounce less than one path per thread.
// Which scenes: the path integrator on scenes whose tables fit LDS with <= 4 emitting lights (every scene of the reference),
// traversal modes 2 (flat leaf list), 0 / 3 (binary + 8-wide trees, walkers) and 5 (reference semantics).  Mode 1 (a small tree
// staged into LDS next to its stack) and larger tables keep the per-bounce launches.
bool fused_eligible(const JpContext* c, const JpRenderParams* rp)
{
	const char* e = getenv("JETPBRT_FUSED");
	if (!e || atoi(e) == 0) return false;
	if (c->is_lane || !c->have_scene || rp->integrator != JP_INTEGRATOR_PATH) return false;
	if (!c->tables_in_lds || c->n_planes > 4) return false;          // (its own LDS budget: render_fused shrinks the region until the layout fits)
	if (c->cert) return false;                                       // the certified walk lives in the per-bounce traversal kernels
	if (c->trav_mode == 2) return c->shade_prims_in_lds;
	return c->trav_mode == 0 || c->trav_mode == 3 || c->trav_mode == 5;
}

typedef void (*PathKernel)(SceneView, Queues, RenderConst, PathConst, int*, DevCounters*);
PathKernel path_kernel(const JpContext* c)
{
	const bool so = c->shade_sort;
	switch (c->trav_mode)
	{
	case 2: return so ? k_path<2, 2, true, true, true> : k_path<2, 2, true, false, true>;
	case 3:
		if (c->use_q4) return c->q4_shadow ? (so ? k_path<4, 4, false, true, true> : k_path<4, 4, false, false, true>) : (so ? k_path<4, 3, false, true, true> : k_path<4, 3, false, false, true>);
		return so ? k_path<0, 3, false, true, true> : k_path<0, 3, false, false, true>;
	case 5: return so ? k_path<5, 5, false, true, false> : k_path<5, 5, false, false, false>;
	default:
		if (c->use_q4) return so ? k_path<4, 4, false, true, true> : k_path<4, 4, false, false, true>;
		return so ? k_path<0, 0, false, true, true> : k_path<0, 0, false, false, true>;
	}
}

int render_fused(JpContext* c, const JpRenderParams* rp, float* film_dev, bool sync)
{
	if (rp->width <= 0 || rp->height <= 0 || rp->spp <= 0 || rp->max_depth < 0 || rp->max_depth > 200) return fail(JP_ERR_INVALID_ARGUMENT, "jp_render: bad width/height/spp/max_depth");
	if (rp->sampler_mode != JP_SAMPLER_COUNTER && rp->sampler_mode != JP_SAMPLER_DEBUG) return fail(JP_ERR_UNSUPPORTED, "jp_render: the device path implements the counter sampler only (the sequential mt19937_64 stream is not reproducible in parallel)");
	const int band = rp->band_rows > 0 ? rp->band_rows : 20;
	const int scount = rp->shard_count > 1 ? rp->shard_count : 1;
	const int sidx = scount > 1 ? rp->shard_index : 0;
	if (sidx < 0 || sidx >= scount) return fail(JP_ERR_INVALID_ARGUMENT, "jp_render: shard_index out of range");
	HIP_TRY(hipSetDevice(c->device));
	const int nbands = (rp->height + band - 1) / band;
	int local_rows = 0;
	for (int b = sidx; b < nbands; b += scount) local_rows += std::min(band, rp->height - b * band);
	const long long npix = (long long)local_rows * rp->width;

	HIP_TRY(hipEventRecord(c->ev0, c->stream));
	HIP_TRY(hipMemsetAsync(film_dev, 0, sizeof(float) * 3 * (size_t)rp->width * rp->height, c->stream));
	HIP_TRY(hipMemsetAsync(c->d_cnt, 0, sizeof(DevCounters), c->stream));
	c->evused = 0; c->stamps.clear();
	unsigned long long samples = 0;
	c->last_fused = 1; c->last_lanes = 1;
	if (npix > 0)
	{
		if (npix > (1 << 24)) return fail(JP_ERR_UNSUPPORTED, "jp_render: more than 2^24 pixels per shard");
		const PathKernel kern = path_kernel(c);
		const bool flat = c->trav_mode == 2;
		const int modeE = flat ? 2 : (c->trav_mode == 5 ? 5 : 0);
		// ---- batch: the radiance array holds one float4 per path of the batch (the only per-path array that outlives a job) ----
		size_t freeB = 0, totalB = 0; hipMemGetInfo(&freeB, &totalB);
		size_t budget = std::min<size_t>((size_t)4 << 30, (freeB + c->flacc_n * 16) / 4);
		if (const char* e = getenv("JETPBRT_MAX_SLOTS")) { long long v = atoll(e); if (v > 0) budget = std::min<size_t>(budget, (size_t)v * 16); }
		const size_t PMAX = (size_t)1 << 26;
		const size_t pcap = std::min<size_t>(PMAX, std::max<size_t>((size_t)npix, budget / 16));
		int sbatch = (int)std::max<long long>(1, std::min<long long>(rp->spp, (long long)(pcap / (size_t)npix)));
		{ const int nb = (rp->spp + sbatch - 1) / sbatch; sbatch = (rp->spp + nb - 1) / nb; }       // equal batches
		// ---- job shape: R paths = PG pixels x S samples.  A wave's 64 lanes are 64 neighbouring pixels of one sample. ----
		unsigned int R = 1024;
		if (const char* e = getenv("JETPBRT_REGION")) { int v = atoi(e); if (v >= JP_BLOCK && v <= 8192) R = (unsigned int)(v / JP_BLOCK) * JP_BLOCK; }
		int S = 16;
		if (const char* e = getenv("JETPBRT_JOB_SPP")) { int v = atoi(e); if (v >= 1 && v <= 128) S = v; }
		const int n_tab = 2 * c->sv.n_lights + 4 * c->sv.n_mats + (c->sv.n_mats + 3) / 4, n_tab_all = n_tab + (flat ? 8 * c->sv.n_prims : 0);
		const int deepE = c->stack_depth, deepS = (c->trav_mode == 3 && !c->q4_shadow) ? (int)(c->lds_bytes_shadow / (JP_BLOCK * sizeof(int))) : c->stack_depth;
		const int ecap = flat ? 0 : std::min(deepE, c->stack_lds_words), scap = flat ? 0 : std::min(deepS, c->stack_lds_words);
		PathLds L = path_lds_layout(modeE, n_tab_all, c->sv.n_prims, R, c->n_planes, ecap, scap, c->shade_sort);
		while (L.total > 64 * 1024 && R > JP_BLOCK) { R -= JP_BLOCK; L = path_lds_layout(modeE, n_tab_all, c->sv.n_prims, R, c->n_planes, ecap, scap, c->shade_sort); }
		if (L.total > 64 * 1024) return fail(JP_ERR_UNSUPPORTED, "jp_render: the fused schedule's LDS layout does not fit this scene (JETPBRT_FUSED=0 selects the per-bounce launches)");
		S = std::max(1, std::min(S, std::min(sbatch, (int)(R / 64))));
		int PG = (int)(R / (unsigned int)S); if (PG >= 64) PG &= ~63;
		if ((long long)PG > npix) PG = (int)npix;
		const int npg = (int)((npix + PG - 1) / PG), nsb = (sbatch + S - 1) / S;
		const unsigned long long njobs = (unsigned long long)npg * nsb;
		if (njobs >= (1ull << 32)) return fail(JP_ERR_UNSUPPORTED, "jp_render: too many jobs per batch");
		// ---- resident workgroups: as many as the kernel's registers and LDS allow, persistent, taking jobs from a counter ----
		HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.total));
		int per_cu = 0;
		HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kern, JP_BLOCK, L.total));
		per_cu = std::max(1, per_cu);
		if (const char* e = getenv("JETPBRT_FUSED_WGS")) { int v = atoi(e); if (v >= 1 && v <= 16) per_cu = v; }
		const unsigned int G = (unsigned int)std::min<unsigned long long>(njobs, (unsigned long long)c->n_cus * per_cu);
		const unsigned int cap = G * R;
		if (c->fcap < cap || c->fplanes < c->n_planes)
		{
			HIP_TRY(hipStreamSynchronize(c->stream));
			const unsigned int ncap = std::max(cap, c->fcap); const int npl = std::max(c->n_planes, c->fplanes);
			const size_t keep_lacc = c->flacc_n; float4* keep = c->fq.lacc;
			for (void* p : c->fbufs) if (p != (void*)keep) hipFree(p);
			c->fbufs.clear(); if (keep) c->fbufs.push_back(keep);
			c->fcap = 0; c->fplanes = 0;
			Queues& q = c->fq; float4* lacc = keep; std::memset(&q, 0, sizeof(q)); q.lacc = lacc; c->flacc_n = keep_lacc;
			auto alloc = [&](void** p, size_t bytes) -> bool { if (hipMalloc(p, bytes) != hipSuccess) return false; c->fbufs.push_back(*p); return true; };
			bool ok = true;
			for (int b = 0; b < 2 && ok; b++) ok = alloc((void**)&q.ray_o[b], (size_t)ncap * 16) && alloc((void**)&q.ray_d[b], (size_t)ncap * 16) && alloc((void**)&q.beta[b], (size_t)ncap * 16);
			ok = ok && alloc((void**)&q.sh_o, (size_t)ncap * 16) && alloc((void**)&q.sh_d, (size_t)ncap * 16 * npl) && alloc((void**)&q.sh_c, (size_t)ncap * 16 * npl);
			if (!ok) { free_fused(c); return fail(JP_ERR_DEVICE, "jp_render: out of device memory for the region queues"); }
			c->fcap = ncap; c->fplanes = npl;
		}
		const size_t P = (size_t)sbatch * (size_t)npix;
		if (c->flacc_n < P)
		{
			HIP_TRY(hipStreamSynchronize(c->stream));
			if (c->fq.lacc) { c->fbufs.erase(std::remove(c->fbufs.begin(), c->fbufs.end(), (void*)c->fq.lacc), c->fbufs.end()); hipFree(c->fq.lacc); c->fq.lacc = nullptr; c->flacc_n = 0; }
			void* pl = nullptr; if (hipMalloc(&pl, P * 16) != hipSuccess) return fail(JP_ERR_DEVICE, "jp_render: out of device memory for the batch's radiance array");
			c->fq.lacc = (float4*)pl; c->fbufs.push_back(pl); c->flacc_n = P;
		}
		c->fq.cap = c->fcap; c->fq.R = R;
		const int nbatches = (rp->spp + sbatch - 1) / sbatch;
		if (c->jobs_n < (size_t)nbatches) { HIP_TRY(hipStreamSynchronize(c->stream)); if (c->d_jobs) hipFree(c->d_jobs); c->d_jobs = nullptr; c->jobs_n = 0; HIP_TRY(hipMalloc((void**)&c->d_jobs, (size_t)nbatches * 4)); c->jobs_n = (size_t)nbatches; }
		HIP_TRY(hipMemsetAsync(c->d_jobs, 0, (size_t)nbatches * 4, c->stream));
		{   // spill area of the walkers' stacks beyond the words kept in LDS
			const int deep = std::max(deepE, deepS);
			const size_t need = !flat && deep >= c->stack_lds_words ? (size_t)(deep - c->stack_lds_words + 1) * G * JP_BLOCK : 1;   // (+1: Walker<4>'s dump slot)
			if (c->spill_words < need) { HIP_TRY(hipStreamSynchronize(c->stream)); if (c->d_spill) hipFree(c->d_spill); c->d_spill = nullptr; c->spill_words = 0; HIP_TRY(hipMalloc((void**)&c->d_spill, need * sizeof(int))); c->spill_words = need; }
		}
		if (c->pix_acc_n < (size_t)npix) { HIP_TRY(hipStreamSynchronize(c->stream)); if (c->d_pix_acc) hipFree(c->d_pix_acc); c->d_pix_acc = nullptr; HIP_TRY(hipMalloc((void**)&c->d_pix_acc, (size_t)npix * 16)); c->pix_acc_n = (size_t)npix; }

		RenderConst rc; std::memset(&rc, 0, sizeof(rc));
		rc.width = rp->width; rc.height = rp->height; rc.spp = rp->spp; rc.max_depth = rp->max_depth; rc.seed = rp->seed;
		rc.band_rows = band; rc.shard_index = sidx; rc.shard_count = scount; rc.npix = (int)npix; rc.local_rows = local_rows; rc.n_planes = c->n_planes;
		rc.lane_index = 0; rc.lane_count = 1; rc.lane_rows = 4; rc.class_mask = c->class_mask; rc.sampler_debug = rp->sampler_mode == JP_SAMPLER_DEBUG ? 1 : 0;
		rc.slot_bits = JP_PATH_LI_BITS;
		// 16 x 4 pixel tiles: a job's 64-pixel groups are patches of the image, so the lanes of a wave start as neighbours (camera
		// rays and first shadow rays of large scenes share nodes).  JETPBRT_NO_TILES=1: row-major groups.
		rc.tiled = (rp->width % 16 == 0 && local_rows % 4 == 0 && PG % 64 == 0 && !getenv("JETPBRT_NO_TILES") && (c->trav_mode != 2 || getenv("JETPBRT_TILES"))) ? 1 : 0;
		PathConst pc; pc.R = R; pc.PG = PG; pc.S = S; pc.npg = npg; pc.nsb = nsb; pc.ecap = ecap; pc.scap = scap;
		pc.max_iters = rp->max_depth + 1 + (c->has_null_material ? 64 : 0);
		c->last_region = (int)R; c->last_wgs = (int)G;
		for (int s0 = 0, bi = 0; s0 < rp->spp; s0 += sbatch, bi++)
		{
			rc.s0 = s0; rc.sbatch = std::min(sbatch, rp->spp - s0);
			pc.nsb = (rc.sbatch + S - 1) / S; pc.job = c->d_jobs + bi;
			const unsigned int g = (unsigned int)std::min<unsigned long long>((unsigned long long)npg * pc.nsb, (unsigned long long)G);
			{ Stamper t(c, CLS_PATH); hipLaunchKernelGGL(kern, dim3(g), dim3(JP_BLOCK), L.total, c->stream, c->sv, c->fq, rc, pc, c->d_spill, c->d_cnt); }
			HIP_TRY(hipGetLastError());
			{ Stamper t(c, CLS_OTHER); hipLaunchKernelGGL(k_resolve, dim3((unsigned int)std::min<long long>(c->n_cus * 8, (npix + JP_BLOCK - 1) / JP_BLOCK)), dim3(JP_BLOCK), 0, c->stream, c->fq, rc, c->d_pix_acc, film_dev, s0 == 0 ? 1 : 0, s0 + rc.sbatch >= rp->spp ? 1 : 0); }
			HIP_TRY(hipGetLastError());
			samples += (unsigned long long)rc.sbatch * (unsigned long long)npix;
		}
	}
	HIP_TRY(hipEventRecord(c->ev1, c->stream));
	c->own_samples = samples;
	if (sync) HIP_TRY(hipStreamSynchronize(c->stream));
	return JP_OK;
}

int render_impl(JpContext* c, const JpRenderParams* rp, float* film_dev, bool sync)
{
	if (!c || !rp || !film_dev) return fail(JP_ERR_INVALID_ARGUMENT, "jp_render: null argument");
	c->last_lanes = 1; c->last_fused = 0;
	if (fused_eligible(c, rp)) return render_fused(c, rp, film_dev, sync);
	// lanes: the shard's rows in groups of 4 dealt round-robin to L contexts.  Default: 3 lanes when each gets >= 16 groups and
	// full-size batches, else 2, else 1 (measured on the benchmark frame: 1 lane 2.19, 2 lanes 2.70, 3 lanes 2.82, 4 lanes 2.38
	// Gsamples/s).  JETPBRT_LANES = 1 .. 4 forces a count, JETPBRT_LANE_ROWS the group height.
	int forcedL = 0, group = 4;
	if (const char* e = getenv("JETPBRT_LANES")) { int v = atoi(e); if (v >= 1 && v <= 4) forcedL = v; }
	if (const char* e = getenv("JETPBRT_LANE_ROWS")) { int v = atoi(e); if (v >= 1 && v <= 64) group = v; }
	int L = 1;
	const int band = rp->band_rows > 0 ? rp->band_rows : 20;
	const int scount = rp->shard_count > 1 ? rp->shard_count : 1;
	const int sidx = scount > 1 ? rp->shard_index : 0;
	if (!c->is_lane && c->have_scene && !c->has_null_material && rp->width > 0 && rp->height > 0 && rp->integrator == JP_INTEGRATOR_PATH && sidx >= 0 && sidx < scount)
	{
		const int nbands = (rp->height + band - 1) / band;
		long long rows = 0;
		for (int b = sidx; b < nbands; b += scount) rows += std::min(band, rp->height - b * band);
		const long long groups = (rows + group - 1) / group;
		if (forcedL) L = (int)std::min<long long>(forcedL, std::max<long long>(1, groups));
		else
		{
			// worth it only when each lane still gets full-size batches (2^24 slots): measured -7 % at 512 x 512 x 64 spp
			// (half-size batches), +17 % / +24 % at 1024 spp with two / three lanes
			const long long samples = rows * rp->width * (long long)rp->spp;
			// round 2, measured on one rank's share of an 8- / 4-GPU frame (64 / 128 rows of 512 x 512, tools/gpu_shard_lanes.py): three lanes
			// beat two there as well (1/8 shard at 1024 spp 13.5 vs 14.8 ms, at 8192 spp 2570 vs 2334 Msamples/s), so the lane count follows
			// the sample count alone
			if (groups >= 3 && samples >= (2ll << 24)) L = 3;
			else if (groups >= 2 && samples >= (2ll << 24)) L = 2;
		}
	}
	if (L <= 1) return render_one(c, rp, film_dev, sync);

	HIP_TRY(hipSetDevice(c->device));
	int st = make_lanes(c, L - 1); if (st != JP_OK) return st;
	const size_t n = (size_t)rp->width * rp->height * 3;
	// workgroups per CU and lane (measured on the benchmark frame, two lanes: 2.51 Gsamples/s at 16 + 16, 2.70 at 8 + 8,
	// 2.74 at 6 + 6, 2.60 at 4 + 4; three lanes: 2.83 at 5 + 5 + 5; a single lane is best at 16)
	HIP_TRY(hipEventRecord(c->ev0, c->stream));                                       // render_ms starts before the first lane's kernels are enqueued
	const int bpc_single = c->blocks_per_cu, bpc_lane = c->bpc_from_env ? c->blocks_per_cu : std::max(4, 16 / L);
	for (int k = 1; k < L && st == JP_OK; k++)
	{
		JpContext* l = c->lanes[k - 1];
		sync_lane_scene(c, l);
		if (l->film_n < n) { if (l->d_film) { HIP_TRY(hipStreamSynchronize(c->stream)); hipFree(l->d_film); } l->d_film = nullptr; l->film_n = 0; HIP_TRY(hipMalloc((void**)&l->d_film, n * sizeof(float))); l->film_n = n; c->added_valid = false; }
		if (c->added_valid) HIP_TRY(hipStreamWaitEvent(l->stream, c->ev_added, 0));   // the previous frame's merge still reads the lane film
		l->blocks_per_cu = bpc_lane;
		st = render_one(l, rp, l->d_film, false, k, L, group);
	}
	if (st == JP_OK) { c->blocks_per_cu = bpc_lane; st = render_one(c, rp, film_dev, false, 0, L, group, true); c->blocks_per_cu = bpc_single; }
	if (st != JP_OK) return st;
	for (int k = 1; k < L; k++)
	{
		JpContext* l = c->lanes[k - 1];
		HIP_TRY(hipStreamWaitEvent(c->stream, l->ev1, 0));                           // recorded at the end of the lane's render_one
		hipLaunchKernelGGL(k_add_film, dim3((unsigned int)std::min<size_t>((size_t)c->n_cus * 8, (n + JP_BLOCK - 1) / JP_BLOCK)), dim3(JP_BLOCK), 0, c->stream, film_dev, (const float*)l->d_film, n);
	}
	HIP_TRY(hipEventRecord(c->ev_added, c->stream)); c->added_valid = true;
	HIP_TRY(hipEventRecord(c->ev1, c->stream));                                       // render_ms: all lanes and the merge
	c->last_lanes = L;
	if (sync) HIP_TRY(hipStreamSynchronize(c->stream));
	return JP_OK;
}

int finish_one(JpContext* c, JpCounters& o)
{
	HIP_TRY(hipStreamSynchronize(c->stream));
	DevCounters h; HIP_TRY(hipMemcpy(&h, c->d_cnt, sizeof(h), hipMemcpyDeviceToHost));
	o.closest_rays += h.closest; o.closest_hits += h.closest_hit; o.shadow_rays += h.shadow; o.shadow_occluded += h.shadow_occ; o.certified_fallback_rays += h.cert_fallback;
	for (const JpContext::Stamp& s : c->stamps)
	{
		float t = 0.f; if (hipEventElapsedTime(&t, c->evpool[s.a], c->evpool[s.b]) != hipSuccess) continue;
		if (s.cls == CLS_EXTEND) { o.extend_ms += t; o.extend_launches++; }
		else if (s.cls == CLS_SHADE) { o.shade_ms += t; o.shade_launches++; }
		else if (s.cls == CLS_SHADOW) { o.shadow_ms += t; o.shadow_launches++; }
		else if (s.cls == CLS_PATH) { o.path_ms += t; o.path_launches++; }
		else o.other_ms += t;
	}
	return JP_OK;
}

int finish_counters(JpContext* c)
{
	HIP_TRY(hipSetDevice(c->device));
	JpCounters& o = c->counters;
	unsigned long long samples = c->own_samples;
	std::memset(&o, 0, sizeof(o));
	int st = finish_one(c, o); if (st != JP_OK) return st;
	for (int k = 1; k < c->last_lanes; k++)                                           // per-class times add up over the (overlapping) lanes
	{ samples += c->lanes[k - 1]->own_samples; st = finish_one(c->lanes[k - 1], o); if (st != JP_OK) return st; }
	float ms = 0.f; if (hipEventElapsedTime(&ms, c->ev0, c->ev1) != hipSuccess) ms = 0.f;
	o.render_ms = ms; o.samples = samples;
	return JP_OK;
}
}

extern "C" {

int jp_render_device(JpContext* c, const JpRenderParams* rp, void* film_rgb_device, int sync) { return render_impl(c, rp, (float*)film_rgb_device, sync != 0); }

int jp_render(JpContext* c, const JpRenderParams* rp, float* film_host)
{
	if (!c || !rp || !film_host) return fail(JP_ERR_INVALID_ARGUMENT, "jp_render: null argument");
	if (rp->width <= 0 || rp->height <= 0) return fail(JP_ERR_INVALID_ARGUMENT, "jp_render: bad width/height");
	HIP_TRY(hipSetDevice(c->device));
	size_t n = (size_t)rp->width * rp->height * 3;
	if (c->film_n < n) { if (c->d_film) hipFree(c->d_film); c->d_film = nullptr; HIP_TRY(hipMalloc((void**)&c->d_film, n * sizeof(float))); c->film_n = n; }
	if (c->h_film_n < n) { if (c->h_film) hipHostFree(c->h_film); c->h_film = nullptr; c->h_film_n = 0; if (hipHostMalloc((void**)&c->h_film, n * sizeof(float), hipHostMallocDefault) == hipSuccess) c->h_film_n = n; else c->h_film = nullptr; }
	int st = render_impl(c, rp, c->d_film, false); if (st != JP_OK) return st;
	float* stage = c->h_film ? c->h_film : film_host;
	HIP_TRY(hipMemcpyAsync(stage, c->d_film, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));
	if (stage != film_host) std::memcpy(film_host, stage, n * sizeof(float));
	return JP_OK;
}

int jp_render_rgb8(JpContext* c, const JpRenderParams* rp, uint8_t* rgb8_host, float* film_host)
{
	if (!c || !rp || !rgb8_host) return fail(JP_ERR_INVALID_ARGUMENT, "jp_render_rgb8: null argument");
	if (rp->width <= 0 || rp->height <= 0) return fail(JP_ERR_INVALID_ARGUMENT, "jp_render_rgb8: bad width/height");
	HIP_TRY(hipSetDevice(c->device));
	const size_t n = (size_t)rp->width * rp->height * 3;
	if (c->film_n < n) { if (c->d_film) hipFree(c->d_film); c->d_film = nullptr; HIP_TRY(hipMalloc((void**)&c->d_film, n * sizeof(float))); c->film_n = n; }
	if (c->rgb8_n < n) { if (c->d_rgb8) hipFree(c->d_rgb8); c->d_rgb8 = nullptr; c->rgb8_n = 0; HIP_TRY(hipMalloc((void**)&c->d_rgb8, n)); c->rgb8_n = n; }
	if (c->h_rgb8_n < n) { if (c->h_rgb8) hipHostFree(c->h_rgb8); c->h_rgb8 = nullptr; c->h_rgb8_n = 0; if (hipHostMalloc((void**)&c->h_rgb8, n, hipHostMallocDefault) == hipSuccess) c->h_rgb8_n = n; else c->h_rgb8 = nullptr; }
	if (!c->d_gamma) { HIP_TRY(hipMalloc((void**)&c->d_gamma, 255 * sizeof(float))); HIP_TRY(hipMemcpy(c->d_gamma, host_gamma_thresholds(), 255 * sizeof(float), hipMemcpyHostToDevice)); }
	int st = render_impl(c, rp, c->d_film, false); if (st != JP_OK) return st;
	hipLaunchKernelGGL(k_tonemap8, dim3((unsigned int)std::min<size_t>((size_t)c->n_cus * 8, (n + JP_BLOCK - 1) / JP_BLOCK)), dim3(JP_BLOCK), 0, c->stream, (const float*)c->d_film, c->d_rgb8, (const float*)c->d_gamma, n);
	unsigned char* stage = c->h_rgb8 ? c->h_rgb8 : rgb8_host;
	HIP_TRY(hipMemcpyAsync(stage, c->d_rgb8, n, hipMemcpyDeviceToHost, c->stream));
	if (film_host) HIP_TRY(hipMemcpyAsync(film_host, c->d_film, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));
	if (stage != rgb8_host) std::memcpy(rgb8_host, stage, n);
	return JP_OK;
}

int jp_synchronize(JpContext* c) { if (!c) return fail(JP_ERR_INVALID_ARGUMENT, "jp_synchronize: null context"); HIP_TRY(hipSetDevice(c->device)); HIP_TRY(hipStreamSynchronize(c->stream)); return JP_OK; }
int jp_set_profiling(JpContext* c, int enabled) { if (!c) return fail(JP_ERR_INVALID_ARGUMENT, "jp_set_profiling: null context"); c->profiling = enabled != 0; return JP_OK; }
int jp_get_counters(JpContext* c, JpCounters* out)
{
	if (!c || !out) return fail(JP_ERR_INVALID_ARGUMENT, "jp_get_counters: null argument");
	int st = finish_counters(c); if (st != JP_OK) return st;
	*out = c->counters; return JP_OK;
}
int jp_get_build_info(JpContext* c, JpBuildInfo* out)
{
	if (!c || !out) return fail(JP_ERR_INVALID_ARGUMENT, "jp_get_build_info: null argument");
	if (!c->have_scene) return fail(JP_ERR_NO_SCENE, "jp_get_build_info: no scene uploaded");
	out->built_on_device = c->build_on_device ? 1 : 0; out->traversal_mode = c->trav_mode; out->bvh_nodes = c->bvh_nodes; out->bvh_height = c->bvh_height;
	out->device_build_ms = c->build_ms; out->libm_sincosf = c->sincosf_mode; out->lanes_last_render = c->last_lanes;
	out->fused_last_render = c->last_fused; out->fused_region = c->last_region; out->fused_workgroups = c->last_wgs;
	out->q4_nodes = c->use_q4 ? c->sv.n_q4 : 0; out->libm_xbsdf = c->libm_mode;
	out->certified_walk = c->cert ? 1 : 0; out->certified_nodes = c->cert ? c->sv.n_q4 : 0; out->certified_eye_leaves = c->cert ? c->cert_eye_leaves : 0;
	return JP_OK;
}

int jp_bsdf(JpContext* c, const JpBsdfDesc* d, int32_t n, const float* normal, const float* wo, const float* wi, const float* u,
            float* f_eval, float* pdf_eval, float* s_f, float* s_wi, float* s_pdf, int32_t* s_flags)
{
	if (!c || !d || n < 0 || !normal || !wo || !wi || !u || !f_eval || !pdf_eval || !s_f || !s_wi || !s_pdf || !s_flags) return fail(JP_ERR_INVALID_ARGUMENT, "jp_bsdf: null argument");
	if (d->kind < JP_BSDF_LAMBERT || d->kind > JP_BSDF_PHONG) return fail(JP_ERR_INVALID_ARGUMENT, "jp_bsdf: unknown BSDF kind");
	if ((d->kind == JP_BSDF_MICROFACET_REFLECTION || d->kind == JP_BSDF_MICROFACET_TRANSMISSION) && (d->distribution < JP_DIST_TROWBRIDGE_REITZ || d->distribution > JP_DIST_BECKMANN))
		return fail(JP_ERR_INVALID_ARGUMENT, "jp_bsdf: unknown microfacet distribution");
	if (d->kind == JP_BSDF_MICROFACET_REFLECTION && (d->fresnel < JP_FRESNEL_CONDUCTOR || d->fresnel > JP_FRESNEL_NOOP)) return fail(JP_ERR_INVALID_ARGUMENT, "jp_bsdf: unknown Fresnel term");
	if (d->kind == JP_BSDF_FRESNEL_SPECULAR && d->eta_a != 1.0f) return fail(JP_ERR_UNSUPPORTED, "jp_bsdf: FFresnelSpecular is implemented for etaI = 1 (FGlassMaterial, material.h:72-75)");
	if (n == 0) return JP_OK;
	HIP_TRY(hipSetDevice(c->device));
	// scratch buffers kept in the context (a host FBSDF::Evalf is one event per call: no allocation per event); every copy checked
	if (c->bsdf_cap < (size_t)n)
	{
		HIP_TRY(hipStreamSynchronize(c->stream));
		if (c->d_bsdf_in) hipFree(c->d_bsdf_in); if (c->d_bsdf_out) hipFree(c->d_bsdf_out); if (c->d_bsdf_fl) hipFree(c->d_bsdf_fl);
		c->d_bsdf_in = c->d_bsdf_out = nullptr; c->d_bsdf_fl = nullptr; c->bsdf_cap = 0;
		const size_t cap = std::max<size_t>((size_t)n, 256);
		if (hipMalloc((void**)&c->d_bsdf_in, cap * 11 * 4) != hipSuccess || hipMalloc((void**)&c->d_bsdf_out, cap * 11 * 4) != hipSuccess || hipMalloc((void**)&c->d_bsdf_fl, cap * 4) != hipSuccess)
			return fail(JP_ERR_DEVICE, "jp_bsdf: out of device memory");
		c->bsdf_cap = cap;
	}
	float *dn = c->d_bsdf_in, *dwo = dn + 3 * (size_t)n, *dwi = dn + 6 * (size_t)n, *du = dn + 9 * (size_t)n;
	float *df = c->d_bsdf_out, *dpe = df + 3 * (size_t)n, *dsf = df + 4 * (size_t)n, *dswi = df + 7 * (size_t)n, *dsp = df + 10 * (size_t)n;
	int* dfl = c->d_bsdf_fl;
	HIP_TRY(hipMemcpyAsync(dn, normal, (size_t)n * 12, hipMemcpyHostToDevice, c->stream)); HIP_TRY(hipMemcpyAsync(dwo, wo, (size_t)n * 12, hipMemcpyHostToDevice, c->stream));
	HIP_TRY(hipMemcpyAsync(dwi, wi, (size_t)n * 12, hipMemcpyHostToDevice, c->stream)); HIP_TRY(hipMemcpyAsync(du, u, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
	const int grid = std::min(c->n_cus * 8, (n + JP_BLOCK - 1) / JP_BLOCK);
	hipLaunchKernelGGL(k_bsdf, dim3(grid), dim3(JP_BLOCK), 0, c->stream, *d, n, (const float*)dn, (const float*)dwo, (const float*)dwi, (const float*)du, df, dpe, dsf, dswi, dsp, dfl);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipMemcpyAsync(f_eval, df, (size_t)n * 12, hipMemcpyDeviceToHost, c->stream)); HIP_TRY(hipMemcpyAsync(pdf_eval, dpe, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipMemcpyAsync(s_f, dsf, (size_t)n * 12, hipMemcpyDeviceToHost, c->stream)); HIP_TRY(hipMemcpyAsync(s_wi, dswi, (size_t)n * 12, hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipMemcpyAsync(s_pdf, dsp, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream)); HIP_TRY(hipMemcpyAsync(s_flags, dfl, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));
	return JP_OK;
}

int jp_trace(JpContext* c, int32_t n, const float* origin, const float* dir, const float* tmin, const float* tmax, int32_t* hit, float* t, int32_t* prim, float* normal)
{
	if (!c || n < 0 || !origin || !dir || !tmin || !tmax || !hit || !t || !prim || !normal) return fail(JP_ERR_INVALID_ARGUMENT, "jp_trace: null argument");
	if (!c->have_scene) return fail(JP_ERR_NO_SCENE, "jp_trace: no scene uploaded");
	if (n == 0) return JP_OK;
	HIP_TRY(hipSetDevice(c->device));
	float *d_o = nullptr, *d_d = nullptr, *d_t0 = nullptr, *d_t1 = nullptr, *d_t = nullptr, *d_n = nullptr; int *d_hit = nullptr, *d_prim = nullptr;
	int rc = JP_OK;
	do
	{
		if (hipMalloc((void**)&d_o, (size_t)n * 12) != hipSuccess || hipMalloc((void**)&d_d, (size_t)n * 12) != hipSuccess || hipMalloc((void**)&d_t0, (size_t)n * 4) != hipSuccess
		    || hipMalloc((void**)&d_t1, (size_t)n * 4) != hipSuccess || hipMalloc((void**)&d_t, (size_t)n * 4) != hipSuccess || hipMalloc((void**)&d_n, (size_t)n * 12) != hipSuccess
		    || hipMalloc((void**)&d_hit, (size_t)n * 4) != hipSuccess || hipMalloc((void**)&d_prim, (size_t)n * 4) != hipSuccess) { rc = fail(JP_ERR_DEVICE, "jp_trace: out of device memory"); break; }
		hipMemcpyAsync(d_o, origin, (size_t)n * 12, hipMemcpyHostToDevice, c->stream); hipMemcpyAsync(d_d, dir, (size_t)n * 12, hipMemcpyHostToDevice, c->stream);
		hipMemcpyAsync(d_t0, tmin, (size_t)n * 4, hipMemcpyHostToDevice, c->stream); hipMemcpyAsync(d_t1, tmax, (size_t)n * 4, hipMemcpyHostToDevice, c->stream);
		int grid = std::min(c->n_cus * 8, (n + JP_BLOCK - 1) / JP_BLOCK);
		if (c->trav_mode == 3 && getenv("JETPBRT_TRACE_WIDE")) hipLaunchKernelGGL(k_trace<3>, dim3(grid), dim3(JP_BLOCK), c->lds_bytes_shadow, c->stream, c->sv, c->stack_depth, n, d_o, d_d, d_t0, d_t1, d_hit, d_t, d_prim, d_n);
		else if (c->trav_mode == 5 && c->cert && (size_t)c->stack_depth * JP_BLOCK * sizeof(int) <= 64 * 1024 && !getenv("JETPBRT_TRACE_VERBATIM")) hipLaunchKernelGGL(k_trace<6>, dim3(grid), dim3(JP_BLOCK), (size_t)c->stack_depth * JP_BLOCK * sizeof(int), c->stream, c->sv, c->stack_depth, n, d_o, d_d, d_t0, d_t1, d_hit, d_t, d_prim, d_n);
		else if (c->trav_mode == 5) hipLaunchKernelGGL(k_trace<5>, dim3(grid), dim3(JP_BLOCK), c->lds_bytes, c->stream, c->sv, c->stack_depth, n, d_o, d_d, d_t0, d_t1, d_hit, d_t, d_prim, d_n);
		else if (c->use_q4 && !getenv("JETPBRT_TRACE_BINARY")) hipLaunchKernelGGL(k_trace<4>, dim3(grid), dim3(JP_BLOCK), (size_t)c->stack_depth * JP_BLOCK * sizeof(int), c->stream, c->sv, c->stack_depth, n, d_o, d_d, d_t0, d_t1, d_hit, d_t, d_prim, d_n);
		else if (c->trav_mode == 2) hipLaunchKernelGGL(k_trace<2>, dim3(grid), dim3(JP_BLOCK), c->lds_bytes, c->stream, c->sv, c->stack_depth, n, d_o, d_d, d_t0, d_t1, d_hit, d_t, d_prim, d_n);
		else if (c->trav_mode == 1) hipLaunchKernelGGL(k_trace<1>, dim3(grid), dim3(JP_BLOCK), c->lds_bytes, c->stream, c->sv, c->stack_depth, n, d_o, d_d, d_t0, d_t1, d_hit, d_t, d_prim, d_n);
		else hipLaunchKernelGGL(k_trace<0>, dim3(grid), dim3(JP_BLOCK), c->lds_bytes, c->stream, c->sv, c->stack_depth, n, d_o, d_d, d_t0, d_t1, d_hit, d_t, d_prim, d_n);
		hipMemcpyAsync(hit, d_hit, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream); hipMemcpyAsync(t, d_t, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream);
		hipMemcpyAsync(prim, d_prim, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream); hipMemcpyAsync(normal, d_n, (size_t)n * 12, hipMemcpyDeviceToHost, c->stream);
		hipError_t e = hipStreamSynchronize(c->stream);
		if (e != hipSuccess) rc = fail(JP_ERR_DEVICE, std::string("jp_trace: ") + hipGetErrorString(e));
	} while (0);
	hipFree(d_o); hipFree(d_d); hipFree(d_t0); hipFree(d_t1); hipFree(d_t); hipFree(d_n); hipFree(d_hit); hipFree(d_prim);
	return rc;
}

} // extern "C"

#if defined(JP_SHADE_TIMING) || defined(JP_TRAV_TIMING)
extern "C" int jp_dbg_shade_timing(unsigned long long* out16)
{
	unsigned long long z[16] = { 0 };
	if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_shade_t), sizeof(z)) != hipSuccess) return -1;
	if (hipMemcpyToSymbol(HIP_SYMBOL(g_shade_t), z, sizeof(z)) != hipSuccess) return -1;
	return 0;
}
#endif

#ifdef JP_WALK_STATS
extern "C" int jp_dbg_walk_stats(unsigned long long* out8)
{
	unsigned long long z[8] = { 0 };
	if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(jp::g_walk_stats), sizeof(z)) != hipSuccess) return -1;
	if (hipMemcpyToSymbol(HIP_SYMBOL(jp::g_walk_stats), z, sizeof(z)) != hipSuccess) return -1;
	return 0;
}
extern "C" int jp_dbg_turn_stats(unsigned long long* out32)
{
	unsigned long long z[32] = { 0 };
	if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(jp::g_turn_stats), sizeof(z)) != hipSuccess) return -1;
	if (hipMemcpyToSymbol(HIP_SYMBOL(jp::g_turn_stats), z, sizeof(z)) != hipSuccess) return -1;
	return 0;
}
#endif

#ifdef JP_PATH_TIMING
extern "C" int jp_dbg_path_timing(unsigned long long* out16)
{
	unsigned long long z[16] = { 0 };
	if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_path_t), sizeof(z)) != hipSuccess) return -1;
	if (hipMemcpyToSymbol(HIP_SYMBOL(g_path_t), z, sizeof(z)) != hipSuccess) return -1;
	return 0;
}
#endif
