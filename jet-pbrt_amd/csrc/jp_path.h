// jet-pbrt_amd/csrc/jp_path.h -- k_path: the whole bounce loop of FPathIntegratorIteration::Li (integrator.cc:316-403) for one
// queue REGION in one launch.  Included by jp_kernels.hip after RenderConst / Queues / rngf / pixel_of / wave_take.
//
// Round 3 (DESIGN.md "One schedule"): the wavefront formulation of rounds 1-2 is unchanged -- ray generation, closest-hit
// traversal, shading with next-event estimation, shadow traversal, with SoA queues and ballot / popcount compaction between them --
// but a queue region is owned by ONE workgroup from the camera sample to the last bounce, so the phases of a region are separated
// by workgroup barriers instead of kernel boundaries:
//   * 2 launches per batch (k_path, k_resolve) instead of 19, no grid-wide drain between bounces; while one workgroup of a CU
//     shades (queue traffic), its neighbours traverse (vector issue): the overlap the three stream lanes of round 2 bought by
//     time-slicing whole kernels now happens inside every CU;
//   * the workgroups are persistent and take JOBS -- (pixel group, sample block) pairs of <= R paths -- from one atomic counter,
//     so the queue memory is (resident workgroups) x R entries (~160 MB, Infinity-Cache sized) instead of 13 GB per stream lane,
//     and a region's records are read back by the CU that wrote them moments earlier;
//   * per-path data that only ever serves the owning workgroup lives in LDS: the hit record (k_extend -> k_shade, 8 B written +
//     12 B read per segment in HBM before) and the path's radiance sum (32 B read-modify-write per emission / light contribution
//     before); a path's radiance goes to HBM once, when its job ends, and k_resolve sums the samples of a pixel in sample order
//     as before (integrator.cc:102-105).
// Every path computes exactly what it computed in the separate kernels (same closures, same traversal steps, same sums in the
// same order): films are bit-identical to the round-2 launches (tests/test_gpu_parity.py, tools/gpu_ab.py JETPBRT_FUSED=0/1).
#pragma once
#include "jp_common.h"

struct PathConst
{
	unsigned int R;            // region capacity per workgroup (multiple of JP_BLOCK, <= 8192)
	int PG;                    // pixels per job
	int S;                     // samples per job (S * PG <= R)
	int npg;                   // pixel groups of this shard: ceil(npix / PG)
	int nsb;                   // sample blocks of this batch: ceil(sbatch / S)
	int ecap, scap;            // traversal-stack words per thread kept in LDS (closest-hit walker, shadow walker)
	int max_iters;             // bounce iterations per job (max_depth + 1; more for scenes with null materials)
	unsigned int* job;         // job counter, zeroed before the launch
};

// LDS of k_path: tables and per-path state persist over the phases of a job, the scratch area is overlaid by the phase that runs.
struct PathLds { unsigned int tab, trav, lacc, hit, ctr, scratch, total; };
__host__ __device__ inline PathLds path_lds_layout(int modeE, int n_tab_all, int n_prims, unsigned int R, int n_planes, int ecap, int scap, bool sort)
{
	PathLds L; unsigned int o = 0;
	L.tab = o; o += (unsigned int)n_tab_all * 16u;                                   // k_shade's tables (lights | mats | mat types [| prims | meta | frames])
	L.trav = o; if (modeE == 2) o += 5u * (unsigned int)n_prims * 16u;              // tiny scenes: primitive records at the 80-byte stride of flat_prims
	L.lacc = o; o += 3u * R * 4u;                                                    // radiance of the job's paths, three planes
	L.hit = o; o += 2u * R * 4u;                                                     // hit records (t, primitive), two planes
	L.ctr = o; o += 64u;
	L.scratch = o;
	const unsigned int seg = (R / JP_BLOCK) * (JP_BLOCK / 64);
	const unsigned int shade = (unsigned int)n_planes * 2u * JP_BLOCK * 16u + (sort ? R * 2u + R + 6u * seg * 4u : 0u) + 32u;
	const unsigned int ext = modeE == 2 ? 0u : (unsigned int)ecap * JP_BLOCK * 4u;
	const unsigned int shd = modeE == 2 ? 0u : (unsigned int)scap * JP_BLOCK * 4u + ((R * (unsigned int)n_planes + 31u) / 32u) * 4u;
	unsigned int m = shade; if (ext > m) m = ext; if (shd > m) m = shd;
	L.total = o + ((m + 15u) & ~15u);
	return L;
}

// number of set bits of a wave mask below the calling lane (v_mbcnt_lo / v_mbcnt_hi): a lane's rank among the lanes of the mask
__device__ __forceinline__ unsigned int below(unsigned long long m) { return __builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u)); }

// a wave adds the sum of its lanes' counts to an LDS word (statistics; all 64 lanes must be active)
__device__ __forceinline__ void wave_count(unsigned int* ctr, unsigned int v)
{
	for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
	if ((threadIdx.x & 63) == 0 && v) atomicAdd(ctr, v);
}

#ifndef JP_PATH_WAVES
#define JP_PATH_WAVES 3      // waves per SIMD the register allocation of k_path aims at
#endif
#define JP_PATH_LI_BITS 13      // a shadow entry's header: local path index (< 8192) in the low 13 bits, ray count above

#ifdef JP_PATH_TIMING
__device__ unsigned long long g_path_t[16];
#define JP_PT(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pt_acc[i] += t_ - pt_last; pt_last = t_; } while (0)
#else
#define JP_PT(i) do { } while (0)
#endif

// kModeE / kModeS: traversal of closest-hit / shadow rays -- 2 flat leaf list (tiny scenes, primitives in LDS), 0 binary tree,
// 3 8-wide quantised tree, 4 4-wide quantised tree, 5 reference semantics; modes 0 / 3 / 4 / 5 run the resumable walkers with lane
// refill (jp_device.h).
template <int kModeE, int kModeS, bool kPrims, bool kSort, bool kVote>
__global__ void __launch_bounds__(JP_BLOCK, JP_PATH_WAVES) k_path(SceneView sc, Queues q, RenderConst rc, PathConst pc, int* spill, DevCounters* cnt)
{
	static_assert((kModeE == 2) == (kModeS == 2), "k_path: the flat leaf list serves both ray kinds");
	constexpr bool kFlat = kModeE == 2;
	constexpr int kRefill = 16;
	constexpr int kWaves = JP_BLOCK / 64;
	const unsigned int tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	const unsigned int R = pc.R;
	const int NP = rc.n_planes;
	const int n_tab = 2 * sc.n_lights + 4 * sc.n_mats + (sc.n_mats + 3) / 4, n_tab_all = n_tab + (kPrims ? 8 * sc.n_prims : 0);
	const PathLds L = path_lds_layout(kModeE, n_tab_all, sc.n_prims, R, NP, pc.ecap, pc.scap, kSort);
	// ---- LDS carving (byte offsets of path_lds_layout; every area 16-byte aligned) ----
	float4* s_tab = s_dyn + L.tab / 16;
	float4* s_lights = s_tab;
	float4* s_mats = s_lights + 2 * sc.n_lights;
	int* s_mtype = (int*)(s_mats + 4 * sc.n_mats);
	float4* s_prims = s_tab + n_tab;
	int4* s_meta = (int4*)(s_prims + 4 * sc.n_prims);
	const float4* s_frames = s_prims + 5 * sc.n_prims;
	float4* s_trav = s_dyn + L.trav / 16;                           // kFlat: primitive records, 80-byte stride
	float* s_lx = (float*)(s_dyn + L.lacc / 16); float* s_ly = s_lx + R; float* s_lz = s_ly + R;
	float* s_ht = (float*)(s_dyn + L.hit / 16); int* s_hp = (int*)(s_ht + R);
	unsigned int* s_ctr = (unsigned int*)(s_dyn + L.ctr / 16);      // [0] chunk counter, [1] fills (rays | shadow entries << 16), [2] job, [3] n, [4] n_sh, [5] walker ray counter
	float4* s_scr = s_dyn + L.scratch / 16;
	// shade scratch
	float4* s_stage = s_scr + tid;                                  // NEE staging [(2k, 2k+1) * 256 + tid]
	unsigned short* s_idx = (unsigned short*)(s_scr + NP * 2 * JP_BLOCK);
	unsigned char* s_key = (unsigned char*)(s_idx + (kSort ? R : 0));
	unsigned int* s_cnt = (unsigned int*)(s_key + (kSort ? R : 0));
	const unsigned int seg = (R / JP_BLOCK) * kWaves;               // (pass, wave) segments of a full region
	unsigned int* s_wsum = s_cnt + (kSort ? 6 * seg : 0);
	// walker scratch
	unsigned int* s_occ = (unsigned int*)s_scr + pc.scap * JP_BLOCK;

	const float4* lights = (const float4*)s_lights;
	const float4* mats = (const float4*)s_mats;
	const int* mat_type = (const int*)s_mtype;
	const float4* prims = kPrims ? (const float4*)s_prims : sc.prims;
	const int4* meta_t = kPrims ? (const int4*)s_meta : sc.meta;
	const unsigned int rbase = blockIdx.x * R;
#ifdef JP_PATH_TIMING
	unsigned long long pt_acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, pt_last = __builtin_amdgcn_s_memtime();
#endif
	// ---- once per workgroup: tables (and, for tiny scenes, the primitive records of the traversal) into LDS ----
	for (int i = tid; i < n_tab_all; i += JP_BLOCK) s_tab[i] = sc.shade_tab[i];
	if (kFlat) for (int i = tid; i < 4 * sc.n_prims; i += JP_BLOCK) s_trav[5 * (i >> 2) + (i & 3)] = sc.prims[i];
	// ray statistics: per-phase wave sums go to LDS words s_ctr[8..11] (closest, hits, shadow rays, occluded); 64-bit totals in thread 0
	unsigned long long t_closest = 0, t_hit = 0, t_shadow = 0, t_occ = 0;
	if (tid < 4) s_ctr[8 + tid] = 0;
	const unsigned int njobs = (unsigned int)pc.npg * (unsigned int)pc.nsb;

	for (;;)
	{
		__syncthreads();                                             // the previous job is done with the LDS (and the tables are staged)
		if (tid == 0) s_ctr[2] = atomicAdd(pc.job, 1u);
		__syncthreads();
		const unsigned int job = (unsigned int)__builtin_amdgcn_readfirstlane((int)s_ctr[2]);   // uniform by construction; says so to the compiler (scalar registers)
		if (job >= njobs) break;
		// job -> (sample block sb, pixel group g): consecutive jobs take different pixel groups of the same sample block
		const unsigned int sb = job / (unsigned int)pc.npg, g = job - sb * (unsigned int)pc.npg;
		const int np = min(pc.PG, rc.npix - (int)g * pc.PG);         // pixels of this group (the shard's last group can be short)
		const int sfirst = (int)sb * pc.S, ns = min(pc.S, rc.sbatch - sfirst);   // samples of this block
		unsigned int n = (unsigned int)(np * ns);                    // paths of this job: local index li = s_rel * np + p_rel
		// ---- ray generation: FSampler::GetCameraSample sampler.h:148-155, FCamera::GenerateRay camera.h:52-58 ----
		for (unsigned int li = tid; li < n; li += JP_BLOCK)
		{
			const int s_rel = (int)li / np, p_rel = (int)li - s_rel * np;
			const int pix = (int)g * pc.PG + p_rel, s = rc.s0 + sfirst + s_rel;
			int x, y; pixel_of(rc, pix, x, y);
			const uint32_t key = jp_rng_key(rc.seed, (uint32_t)x, (uint32_t)y, (uint32_t)s);
			const float fx = (float)x + rngf(rc, key, 0), fy = (float)y + rngf(rc, key, 1);
			const V3 cam_front = mk(sc.cam.front[0], sc.cam.front[1], sc.cam.front[2]);
			const V3 cam_right = mk(sc.cam.right[0], sc.cam.right[1], sc.cam.right[2]);
			const V3 cam_up = mk(sc.cam.up[0], sc.cam.up[1], sc.cam.up[2]);
			V3 dir = cam_front + cam_right * (fx / sc.cam.res_x - 0.5f) + cam_up * (0.5f - fy / sc.cam.res_y);
			dir = normalize(dir);
			q.ray_o[0][rbase + li] = make_float4(sc.cam.pos[0], sc.cam.pos[1], sc.cam.pos[2], __int_as_float((int)li));
			q.ray_d[0][rbase + li] = make_float4(dir.x, dir.y, dir.z, __int_as_float(MK_FLAGS(0, 0, 2)));
			q.beta[0][rbase + li] = make_float4(1.f, 1.f, 1.f, __int_as_float((int)key));
			s_lx[li] = 0.f; s_ly[li] = 0.f; s_lz[li] = 0.f;
		}
		JP_PT(0);                                                    // [0] job take, ray generation
		int cur = 0;
		for (int it = 0; it < pc.max_iters && n > 0; it++)
		{
			const int nxt = cur ^ 1;
			__syncthreads();                                         // the region's rays are written; the scratch area is free
			if (tid == 0) { s_ctr[0] = 0; s_ctr[1] = 0; s_ctr[5] = 0; t_closest += n; }
			// =========================================================================================================
			// extend: FScene::Intersect (scene.cc:25-33) for the region's n rays; hit records to LDS
			// =========================================================================================================
#ifndef JP_DBG_NO_EXTEND
			if constexpr (kFlat)
			{
				unsigned int c_hit = 0;
				float4 ro = make_float4(0, 0, 0, 0), rd = make_float4(0, 0, 1, 0);
				if (tid < n) { ro = q.ray_o[cur][rbase + tid]; rd = q.ray_d[cur][rbase + tid]; }
				for (unsigned int j = tid; j < n; j += JP_BLOCK)
				{
					const float4 co = ro, cd = rd;
					if (j + JP_BLOCK < n) { ro = q.ray_o[cur][rbase + j + JP_BLOCK]; rd = q.ray_d[cur][rbase + j + JP_BLOCK]; }
					float tmax = JP_INF;                             // FRay defaults geometry.h:399: min_t 0.001, max_t infinity
					const int hit = traverse_flat<false, 5>(sc.flat, sc.n_flat, sc.n_prims, (const float4*)s_trav, xyz(co), xyz(cd), 0.001f, tmax);
					s_ht[j] = tmax; s_hp[j] = hit;
					c_hit += hit >= 0 ? 1u : 0u;
				}
				wave_count(&s_ctr[9], c_hit);
			}
			else
			{   // lane refill (k_extend_persist): a lane that finished its ray takes the next one of the region
				__syncthreads();                                     // s_ctr[5] = 0 is visible
				const WalkStack estack = { (int*)s_scr + tid, spill + blockIdx.x * JP_BLOCK + tid, kModeE == 4 ? pc.ecap - 1 : pc.ecap, gridDim.x * JP_BLOCK };   // (Walker<4>: the last LDS word is its dump slot)
				Walker<kModeE> w; w.done = true; w.hit = -1; w.tmax = JP_INF;
				unsigned int ridx = 0xffffffffu, c_hit = 0;
				bool pool = true;
				for (;;)
				{
					const unsigned long long idle = __ballot(w.done);
					const int nidle = __popcll(idle);
					if (nidle == 64 || (pool && nidle >= kRefill))
					{
						if (w.done && ridx != 0xffffffffu) { s_ht[ridx] = w.tmax; s_hp[ridx] = w.hit; c_hit += w.hit >= 0 ? 1u : 0u; ridx = 0xffffffffu; }
						if (!pool) break;
						const int first = __ffsll((long long)idle) - 1;
						unsigned int start = 0;
						if ((int)lane == first) start = atomicAdd(&s_ctr[5], (unsigned int)nidle);
						start = __shfl(start, first);
						pool = start + (unsigned int)nidle < n;
						if (w.done)
						{
							const unsigned int my = start + below(idle);
							if (my < n)
							{
								const float4 ro = q.ray_o[cur][rbase + my], rd = q.ray_d[cur][rbase + my];
								w.start(xyz(ro), xyz(rd), 0.001f, JP_INF);
								ridx = my;
							}
						}
						if (start >= n && nidle == 64) break;
						continue;
					}
					const int nh = __popcll(__ballot(!w.done && w.heavy())), nl = __popcll(__ballot(!w.done && !w.heavy()));
					const bool heavyTurn = kVote ? nh > nl : w.heavy();
					if (!w.done && (w.heavy() == heavyTurn)) w.template step<false>(sc, estack);
				}
				wave_count(&s_ctr[9], c_hit);
			}
#endif
			JP_PT(1);                                                // [1] extend
			__syncthreads();                                         // hit records complete; walker stacks free
			JP_PT(2);                                                // [2] barrier after extend
			// =========================================================================================================
			// shade: the body of Li after the intersection (integrator.cc:328-399); see k_shade for the commentary
			// =========================================================================================================
			const unsigned int count = n;
			if (kSort)
			{   // stable partition of the region's paths by material class (k_shade): s_idx[sorted position] = queue position
				const unsigned int nctr = JP_SHADE_CLASSES * seg;
				const unsigned int npass = (count + JP_BLOCK - 1) / JP_BLOCK;
				for (unsigned int r = 0; r < npass; r++)
				{
					const unsigned int j = r * JP_BLOCK + tid;
					unsigned int key = JP_SHADE_CLASSES;
					if (j < count)
					{
						const int pi = s_hp[j];
						int m = -1; if (pi >= 0) m = meta_t[pi].y;
						key = m >= 0 ? 1u + (unsigned int)mat_type[m] : 0u;
					}
					s_key[r * JP_BLOCK + tid] = (unsigned char)key;
					unsigned int mine = 0;
					#pragma unroll
					for (int c = 0; c < JP_SHADE_CLASSES; c++)
					{
						unsigned int nc = 0;
						if ((rc.class_mask >> c) & 1) nc = (unsigned int)__popcll(__ballot(key == (unsigned int)c));
						if (lane == (unsigned int)c) mine = nc;
					}
					if (lane < JP_SHADE_CLASSES) s_cnt[lane * seg + r * kWaves + wave] = mine;
				}
				__syncthreads();
				{
					unsigned int v[3], t = 0;
					#pragma unroll
					for (int i = 0; i < 3; i++)
					{
						const unsigned int at = 3 * tid + i;
						v[i] = (at < nctr && (at % seg) / kWaves < npass) ? s_cnt[at] : 0u; t += v[i];
					}
					unsigned int incl = t;
					#pragma unroll
					for (int off = 1; off < 64; off <<= 1) { const unsigned int o = __shfl_up(incl, off); if (lane >= (unsigned int)off) incl += o; }
					if (lane == 63) s_wsum[wave] = incl;
					__syncthreads();
					unsigned int base = incl - t;
					#pragma unroll
					for (int w2 = 0; w2 < kWaves; w2++) if ((unsigned int)w2 < wave) base += s_wsum[w2];
					#pragma unroll
					for (int i = 0; i < 3; i++) { const unsigned int at = 3 * tid + i; if (at < nctr) s_cnt[at] = base; base += v[i]; }
				}
				__syncthreads();
				for (unsigned int r = 0; r < npass; r++)
				{
					const unsigned int key = s_key[r * JP_BLOCK + tid];
					unsigned int pre = 0;
					#pragma unroll
					for (int c = 0; c < JP_SHADE_CLASSES; c++)
					{
						if (!((rc.class_mask >> c) & 1)) continue;
						const unsigned long long m = __ballot(key == (unsigned int)c);
						if (key == (unsigned int)c) pre = below(m);
					}
					if (key < JP_SHADE_CLASSES) s_idx[s_cnt[key * seg + r * kWaves + wave] + pre] = (unsigned short)(r * JP_BLOCK + tid);
				}
				__syncthreads();
			}
			JP_PT(3);                                                // [3] partition
			{
				const unsigned int nch = (count + 63u) >> 6;
				// 64-path chunks taken from an LDS counter, from the end of the sorted order (the expensive classes first).  No software
				// prefetch of the next chunk here (k_shade has one): the records were written by this CU a phase ago and the other
				// workgroups of the CU fill the wait; the 13 registers matter more
				for (;;)
				{
					const unsigned int tk = wave_take(&s_ctr[0], 1u);
					if (tk >= nch) break;
					const unsigned int c0 = (nch - 1u - tk) << 6;
					const bool valid = c0 + lane < count;
					float4 ro = make_float4(0, 0, 0, 0), rd = ro, rb = ro; unsigned int pcur = 0;
					if (valid) { pcur = kSort ? (unsigned int)s_idx[c0 + lane] : c0 + lane; ro = q.ray_o[cur][rbase + pcur]; rd = q.ray_d[cur][rbase + pcur]; rb = q.beta[cur][rbase + pcur]; }
					bool shaded = false, wantNee = false, alive = false;
					V3 o = mk(0, 0, 0), d = mk(0, 0, 1), beta = mk(0, 0, 0), p = mk(0, 0, 0), N = mk(0, 0, 1);
					int li = 0, bounce = 0; bool spec = false; unsigned int dim = 0; uint32_t key = 0;
					Closure c; c.kind = CL_LAMBERT; Frame fr; fr.s = fr.t = fr.n = mk(0, 0, 1);
					if (valid)
					{
						o = xyz(ro); d = xyz(rd); beta = xyz(rb);
						li = __float_as_int(ro.w); key = (uint32_t)__float_as_int(rb.w);
						const int flags = __float_as_int(rd.w);
						bounce = FLAG_BOUNCE(flags); spec = FLAG_SPEC(flags); dim = FLAG_DIM(flags);
						const int pi = s_hp[pcur]; const float ht = s_ht[pcur];
						const bool found = pi >= 0;
						int mat = -1, hitprim = 0; bool nflip = false, tabframe = kPrims;
						V3 Le = splat(0);
						if (found)
						{
							const float4 g3 = prims[4 * pi + 3];
							const int4 meta = meta_t[pi];
							const int type = __float_as_int(g3.w);
							p = o + ht * d;                                                       // ray(distance) geometry.h:412-416
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
							for (int l2 = 0; l2 < sc.n_lights; l2++)
							{
								const float4 l0 = lights[2 * l2];
								if (__float_as_int(l0.w) == JP_LIGHT_ENVIRONMENT && !isblack(xyz(l0)))
								{
									const V3 a = mk(s_lx[li], s_ly[li], s_lz[li]) + cmul(beta, xyz(l0));
									s_lx[li] = a.x; s_ly[li] = a.y; s_lz[li] = a.z;
								}
							}
						}
						if (!isblack(Le))
						{
							const V3 a = mk(s_lx[li], s_ly[li], s_lz[li]) + cmul(beta, Le);       // integrator.cc:331
							s_lx[li] = a.x; s_ly[li] = a.y; s_lz[li] = a.z;
						}
						if (found && bounce < rc.max_depth)                                       // integrator.cc:340-343
						{
							if (mat < 0) alive = true;                                            // integrator.cc:349-353: pass through, same bounce
							else
							{
								float up = 0.f;
								const int mtype = mat_type[mat];
								if (mtype == JP_MAT_PLASTIC) up = rngf(rc, key, dim++);           // material.cc:14
								make_closure(mats, mtype, mat, up, c);
								if (kPrims && tabframe)
								{   // FFrame(normal) geometry.h:345-349 from the host-computed table (k_shade)
									const float4 fn = s_frames[3 * hitprim], fs = s_frames[3 * hitprim + 1], ft = s_frames[3 * hitprim + 2];
									fr.n = nflip ? -xyz(fn) : xyz(fn); fr.s = xyz(fs); fr.t = nflip ? -xyz(ft) : xyz(ft);
								}
								else fr = frame_from_z(N);
								shaded = true;
								wantNee = !is_delta(c);
							}
						}
					}
					// ---- next-event estimation (integrator.cc:357-372) ----
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
							for (int l2 = 0; l2 < sc.n_lights; l2++)
							{
								const unsigned int d0 = dim; dim += 2;                          // the two draws are consumed even when the sample is rejected
								const float4 lrad = lights[2 * l2];
								if (isblack(xyz(lrad))) continue;
								const float ux = rngf(rc, key, d0), uy = rngf(rc, key, d0 + 1);
								LightSample ls = sample_li(sc, prims, lights, l2, p, N, ux, uy);
								if (isblack(ls.Li) || ls.pdf == 0.f) continue;
								const V3 f = eval_local(c, wo, to_local(fr, ls.wi));           // FBSDF::Evalf bsdf.h:284-287
								if (isblack(f)) continue;
								const V3 sdir = __float_as_int(lrad.w) == JP_LIGHT_AREA ? ls.wi : normalize(ls.pos - p);   // FScene::Occluded scene.h:36-47
								const float dist = ls.dist >= 0.f ? ls.dist : len(p - ls.pos);
								const V3 contrib = cmul(cmul(beta, f), ls.Li) * absdot(ls.wi, N) / ls.pdf;   // integrator.cc:369
								if (k < NP)
								{
									s_stage[(2 * k) * JP_BLOCK] = make_float4(sdir.x, sdir.y, sdir.z, dist - 0.001f);
									s_stage[(2 * k + 1) * JP_BLOCK] = make_float4(contrib.x, contrib.y, contrib.z, 0.f);
									k++;
								}
							}
						}
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
					// ---- compaction: the wave takes room for its survivors and its shadow entries with one LDS atomic ----
					unsigned int j = 0, qs = 0;
					const unsigned long long ma = __ballot(alive), ms = __ballot(k > 0);
					if (ma | ms)
					{
						const unsigned int base = wave_take(&s_ctr[1], (unsigned int)__popcll(ma) | ((unsigned int)__popcll(ms) << 16));
						j = rbase + (base & 0xffffu) + below(ma);
						qs = rbase + (base >> 16) + below(ms);
					}
					if (k > 0)
					{
						q.sh_o[qs] = make_float4(p.x, p.y, p.z, __int_as_float(li | (k << JP_PATH_LI_BITS)));
						for (int kk = 0; kk < k; kk++)
						{
							q.sh_d[(size_t)kk * q.cap + qs] = s_stage[(2 * kk) * JP_BLOCK];
							q.sh_c[(size_t)kk * q.cap + qs] = s_stage[(2 * kk + 1) * JP_BLOCK];
						}
					}
					if (alive)
					{
						q.ray_o[nxt][j] = make_float4(p.x, p.y, p.z, __int_as_float(li));         // SpawnRay shape.h:61-64
						q.ray_d[nxt][j] = make_float4(nd.x, nd.y, nd.z, __int_as_float(MK_FLAGS(nbounce, nspec, dim)));
						q.beta[nxt][j] = make_float4(nbeta.x, nbeta.y, nbeta.z, __int_as_float((int)key));
					}
				}
			}
			JP_PT(4);                                                // [4] shade
			__syncthreads();                                         // fills complete, outputs written, staging free
			JP_PT(5);                                                // [5] barrier after shade
			const unsigned int fills = (unsigned int)__builtin_amdgcn_readfirstlane((int)s_ctr[1]);
			n = fills & 0xffffu;
			const unsigned int E = fills >> 16;
			// =========================================================================================================
			// shadow: FScene::Occluded (scene.h:36-47) for the entries' rays; visible contributions added in light order
			// =========================================================================================================
#ifndef JP_DBG_NO_SHADOW
			if (E > 0)
			{
				unsigned int c_shadow = 0, c_occ = 0;
				if constexpr (kFlat)
				{
					float4 so_n = make_float4(0, 0, 0, 0), sd_n = make_float4(0, 0, 1, 0);
					if (tid < E) { so_n = q.sh_o[rbase + tid]; sd_n = q.sh_d[rbase + tid]; }
					for (unsigned int j = tid; j < E; j += JP_BLOCK)
					{
						const unsigned int e = rbase + j;
						const float4 so = so_n; float4 sd = sd_n;
						if (j + JP_BLOCK < E) { so_n = q.sh_o[e + JP_BLOCK]; sd_n = q.sh_d[e + JP_BLOCK]; }
						const int packed = __float_as_int(so.w);
						const int li = packed & ((1 << JP_PATH_LI_BITS) - 1), nr = (int)((unsigned int)packed >> JP_PATH_LI_BITS);
						bool any = false;
						V3 a = mk(s_lx[li], s_ly[li], s_lz[li]);
						for (int k = 0; k < nr; k++)
						{
							const float4 c4 = q.sh_c[(size_t)k * q.cap + e];
							float tmax = sd.w;
							const V3 dir = xyz(sd);
							if (k + 1 < nr) sd = q.sh_d[(size_t)(k + 1) * q.cap + e];
							const int hit = traverse_flat<true, 5>(sc.flat, sc.n_flat, sc.n_prims, (const float4*)s_trav, xyz(so), dir, 0.001f, tmax);
							c_shadow++;
							if (hit >= 0) c_occ++;
							else { a = a + xyz(c4); any = true; }
						}
						if (any) { s_lx[li] = a.x; s_ly[li] = a.y; s_lz[li] = a.z; }
					}
				}
				else
				{   // one SHADOW RAY per lane with lane refill (k_shadow_persist); a ray's verdict is one bit of the LDS bitmap
					const unsigned int total = E * (unsigned int)NP;
					for (unsigned int i = tid; i < (total + 31) / 32; i += JP_BLOCK) s_occ[i] = 0;
					if (tid == 0) s_ctr[5] = 0;
					__syncthreads();
					const WalkStack sstack = { (int*)s_scr + tid, spill + blockIdx.x * JP_BLOCK + tid, kModeS == 4 ? pc.scap - 1 : pc.scap, gridDim.x * JP_BLOCK };
					Walker<kModeS> w; w.done = true; w.hit = -1;
					unsigned int rid = 0xffffffffu;
					bool pool = true;
					for (;;)
					{
						const unsigned long long idle = __ballot(w.done);
						const int nidle = __popcll(idle);
						if (nidle == 64 || (pool && nidle >= kRefill))
						{
							if (w.done && rid != 0xffffffffu) { if (w.hit >= 0) atomicOr(&s_occ[rid >> 5], 1u << (rid & 31u)); rid = 0xffffffffu; }
							if (!pool) break;
							const int first = __ffsll((long long)idle) - 1;
							unsigned int start = 0;
							if ((int)lane == first) start = atomicAdd(&s_ctr[5], (unsigned int)nidle);
							start = __shfl(start, first);
							pool = start + (unsigned int)nidle < total;
							if (w.done)
							{
								const unsigned int my = start + below(idle);
								if (my < total)
								{
									const unsigned int k = my / E, e = my - k * E;
									const float4 so = q.sh_o[rbase + e];
									if (k < ((unsigned int)__float_as_int(so.w) >> JP_PATH_LI_BITS))
									{
										const float4 sd = q.sh_d[(size_t)k * q.cap + rbase + e];
										w.start(xyz(so), xyz(sd), 0.001f, sd.w);
										rid = my;
									}
								}
							}
							if (start >= total && nidle == 64) break;
							continue;
						}
						const int nh = __popcll(__ballot(!w.done && w.heavy())), nl = __popcll(__ballot(!w.done && !w.heavy()));
						const bool heavyTurn = kVote ? nh > nl : w.heavy();
						if (!w.done && (w.heavy() == heavyTurn)) w.template step<true>(sc, sstack);
					}
					__syncthreads();
					for (unsigned int e = tid; e < E; e += JP_BLOCK)
					{
						const int packed = __float_as_int(q.sh_o[rbase + e].w);
						const int li = packed & ((1 << JP_PATH_LI_BITS) - 1); const unsigned int nr = (unsigned int)packed >> JP_PATH_LI_BITS;
						unsigned int vm = 0;
						for (unsigned int k = 0; k < nr; k++) { const unsigned int r = k * E + e; if (!((s_occ[r >> 5] >> (r & 31u)) & 1u)) vm |= 1u << k; }
						c_shadow += nr; c_occ += nr - (unsigned int)__popc(vm);
						if (vm)
						{
							V3 a = mk(s_lx[li], s_ly[li], s_lz[li]);
							for (unsigned int k = 0; k < nr; k++)
								if ((vm >> k) & 1u) { const float4 c4 = q.sh_c[(size_t)k * q.cap + rbase + e]; a = a + xyz(c4); }
							s_lx[li] = a.x; s_ly[li] = a.y; s_lz[li] = a.z;
						}
					}
				}
				wave_count(&s_ctr[10], c_shadow); wave_count(&s_ctr[11], c_occ);
			}
#endif
			JP_PT(6);                                                // [6] shadow
			cur = nxt;
		}
		__syncthreads();                                             // every contribution is in
		if (tid == 0) { t_hit += s_ctr[9]; t_shadow += s_ctr[10]; t_occ += s_ctr[11]; s_ctr[9] = 0; s_ctr[10] = 0; s_ctr[11] = 0; }
		// ---- the job's radiance to HBM, once: lacc[slot], slot = sample * npix + pixel (k_resolve sums a pixel's samples in order) ----
		{
			const unsigned int n0 = (unsigned int)(np * ns);
			for (unsigned int li = tid; li < n0; li += JP_BLOCK)
			{
				const int s_rel = (int)li / np, p_rel = (int)li - s_rel * np;
				const size_t slot = (size_t)(sfirst + s_rel) * (size_t)rc.npix + (size_t)((int)g * pc.PG + p_rel);
				q.lacc[slot] = make_float4(s_lx[li], s_ly[li], s_lz[li], 0.f);
			}
		}
		JP_PT(7);                                                    // [7] radiance write-out
	}
	// ---- statistics: one set of atomics per workgroup ----
	if (tid == 0)
	{
		if (t_closest) atomicAdd(&cnt->closest, t_closest);
		if (t_hit) atomicAdd(&cnt->closest_hit, t_hit);
		if (t_shadow) atomicAdd(&cnt->shadow, t_shadow);
		if (t_occ) atomicAdd(&cnt->shadow_occ, t_occ);
	}
#ifdef JP_PATH_TIMING
	if (lane == 0) { for (int i = 0; i < 8; i++) atomicAdd(&g_path_t[i], pt_acc[i]); atomicAdd(&g_path_t[15], 1ull); }
#endif
}
