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
// region, per-iteration vote); k_tonemap8 delivers the
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

// (8 waves per SIMD = 64 VGPRs; the certified walk, which carries the certificate and the verbatim walk behind it: JP_CERT_WAVES)
#ifndef JP_CERT_WAVES
#define JP_CERT_WAVES 8
#endif
template <int kMode, int kRefill, bool kVote>
__global__ void __launch_bounds__(JP_BLOCK, (kMode == 6 ? JP_CERT_WAVES : 8)) k_extend_persist(SceneView sc, Queues q, int cur_q, int stack_cap, int* spill, DevCounters* cnt)
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
__global__ void __launch_bounds__(JP_BLOCK, (kMode == 6 ? JP_CERT_WAVES : 8)) k_shadow_persist(SceneView sc, Queues q, RenderConst rc, int stack_cap, int* spill, DevCounters* cnt)
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

// ---------------------------------------------------------------------------------------------------------------------
// host runtime (same translation unit: the code below launches the kernels above)
// ---------------------------------------------------------------------------------------------------------------------
#include "jp_runtime.h"          // context, options (JpOptions), libm probes, create / destroy
#include "jp_upload.h"           // jp_upload_scene: validation, device tables, trees, device-side build
#include "jp_render.h"           // jp_render*: queues, launch sequence, stream lanes, fused schedule; counters, jp_trace, jp_bsdf
