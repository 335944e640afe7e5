// jet-pbrt_amd/csrc/jp_render.h -- host runtime, part 3 of 3: jp_render* -- queue budget, the per-bounce launch sequence, stream lanes, the fused schedule,
// and the rest of the C ABI (counters, build info, jp_trace, jp_bsdf, tone map).  Included by jp_kernels.hip after jp_upload.h.
#pragma once
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
		if (c->opt.max_slots > 0) budget = std::min<size_t>(budget, (size_t)c->opt.max_slots * per);
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
			const int deep = std::max(std::max(c->stack_depth, c->stack_depth_q4), c->trav_mode == 3 ? (int)(c->lds_bytes_shadow / (JP_BLOCK * sizeof(int))) : 0);
			const size_t need = c->persist && deep >= c->stack_lds_words ? (size_t)(deep - c->stack_lds_words + 1) * G * JP_BLOCK : 1;   // (+1: Walker<4> keeps one LDS word as a dump slot)
			if (c->spill_words < need) { if (c->d_spill) hipFree(c->d_spill); c->d_spill = nullptr; HIP_TRY(hipMalloc((void**)&c->d_spill, need * sizeof(int))); c->spill_words = need; }
		}
		if (c->pix_acc_n < (size_t)npix) { if (c->d_pix_acc) hipFree(c->d_pix_acc); c->d_pix_acc = nullptr; HIP_TRY(hipMalloc((void**)&c->d_pix_acc, (size_t)npix * 16)); c->pix_acc_n = (size_t)npix; }

		RenderConst rc; rc.width = rp->width; rc.height = rp->height; rc.spp = rp->spp; rc.max_depth = rp->max_depth; rc.seed = rp->seed;
		rc.band_rows = band; rc.shard_index = sidx; rc.shard_count = scount; rc.npix = (int)npix; rc.local_rows = local_rows; rc.n_planes = c->n_planes;
		rc.lane_index = lane_index; rc.lane_count = lane_count; rc.lane_rows = lane_group; rc.class_mask = c->class_mask; rc.sampler_debug = rp->sampler_mode == JP_SAMPLER_DEBUG ? 1 : 0;
		// measured: +6 % on the 280k-triangle scene (cache reuse), -8 % on the LDS-resident Cornell box (coherent waves finish
		// together or not at all, which unbalances the workgroups) -> tiles only when traversal goes through global memory
		rc.slot_bits = slot_bits;
		rc.tiled = 0;                                                  // (16 x 4 pixel tiles per wave: measured slower with the wide trees and with the LDS-resident box, profiles/r03e_compact_regions_ab.txt; the fused schedule keeps them)
		// compact regions (k_raygen): scenes walked through global memory -- one lane on the 280k-triangle scene: k_extend 64.1 -> 55.8 ms,
		// k_shadow 47.1 -> 40.5 ms per 512 spp (the workgroups in flight share an image area, hence tree nodes: L2), three lanes +1.2 %;
		// films bit-identical.  JETPBRT_COMPACT_REGIONS=0 / 1 forces it.
		rc.compact = (c->trav_mode != 2 && c->trav_mode != 1 && npix % JP_BLOCK == 0) ? 1 : 0;
		if (c->opt.compact_regions != 0) rc.compact = (c->opt.compact_regions > 0 && npix % JP_BLOCK == 0) ? 1 : 0;
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
			int cur = 0;
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
						const int edepth = (c->trav_mode == 5 ? c->cert : c->use_q4) ? c->stack_depth_q4 : c->stack_depth;
						const int ecap = std::min(edepth, c->stack_lds_words); const size_t elds = (size_t)ecap * JP_BLOCK * sizeof(int);
						#define JP_LAUNCH_EP(M, R) do { if (c->vote) hipLaunchKernelGGL((k_extend_persist<M, R, true>), dim3(grid), dim3(JP_BLOCK), elds, c->stream, c->sv, c->q, cur, ecap, c->d_spill, c->d_cnt); else hipLaunchKernelGGL((k_extend_persist<M, R, false>), dim3(grid), dim3(JP_BLOCK), elds, c->stream, c->sv, c->q, cur, ecap, c->d_spill, c->d_cnt); } while (0)
						if (c->trav_mode == 5 && c->cert) { if (c->persist >= 32) JP_LAUNCH_EP(6, 32); else if (c->persist >= 16) JP_LAUNCH_EP(6, 16); else JP_LAUNCH_EP(6, 8); }
						else if (c->trav_mode == 5) { if (c->persist >= 32) JP_LAUNCH_EP(5, 32); else if (c->persist >= 16) JP_LAUNCH_EP(5, 16); else JP_LAUNCH_EP(5, 8); }
						else if (c->use_q4) { if (c->persist >= 32) JP_LAUNCH_EP(4, 32); else if (c->persist >= 16) JP_LAUNCH_EP(4, 16); else JP_LAUNCH_EP(4, 8); }
						else { if (c->persist >= 32) JP_LAUNCH_EP(0, 32); else if (c->persist >= 16) JP_LAUNCH_EP(0, 16); else JP_LAUNCH_EP(0, 8); }
						#undef JP_LAUNCH_EP
					}
					else if (c->trav_mode == 5) hipLaunchKernelGGL(k_extend<5>, dim3(grid), dim3(JP_BLOCK), lds, c->stream, c->sv, c->q, cur, c->stack_depth, c->d_cnt);
					else if (c->trav_mode == 2) hipLaunchKernelGGL(k_extend<2>, dim3(grid), dim3(JP_BLOCK), lds, c->stream, c->sv, c->q, cur, c->stack_depth, c->d_cnt);
					else if (c->trav_mode == 1) hipLaunchKernelGGL(k_extend<1>, dim3(grid), dim3(JP_BLOCK), lds, c->stream, c->sv, c->q, cur, c->stack_depth, c->d_cnt);
					else hipLaunchKernelGGL(k_extend<0>, dim3(grid), dim3(JP_BLOCK), lds, c->stream, c->sv, c->q, cur, c->stack_depth, c->d_cnt);
				}
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
					hipStream_t sstream = c->stream; int* sspill = c->d_spill;
					Stamper t(c, CLS_SHADOW, sstream);
					const size_t slds = ((c->trav_mode == 5 && c->cert) || (c->trav_mode != 5 && c->q4_shadow)) ? (size_t)c->stack_depth_q4 * JP_BLOCK * sizeof(int) : (c->trav_mode == 3 ? c->lds_bytes_shadow : lds);
					const int scap = std::min((int)(slds / (JP_BLOCK * sizeof(int))), c->stack_lds_words);     // stack words per thread kept in LDS
					const size_t plds = (size_t)scap * JP_BLOCK * sizeof(int) + (((size_t)c->q.R * c->n_planes + 31) / 32) * 4 * (c->cert ? 2 : 1);   // (certified walk: a second bitmap, the rays without a certificate)
					if (c->cert && !(c->persist && plds <= 64 * 1024)) c->cert_fell_back = true;   // the one-ray-per-lane kernels below walk the caller's tree verbatim
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
					else if (c->trav_mode == 3) hipLaunchKernelGGL(k_shadow<3>, dim3(grid), dim3(JP_BLOCK), c->lds_bytes_shadow, c->stream, c->sv, c->q, rc, c->stack_depth, c->d_cnt);
					else if (c->trav_mode == 5) hipLaunchKernelGGL(k_shadow<5>, dim3(grid), dim3(JP_BLOCK), lds, c->stream, c->sv, c->q, rc, c->stack_depth, c->d_cnt);
					else if (c->trav_mode == 2) hipLaunchKernelGGL(k_shadow<2>, dim3(grid), dim3(JP_BLOCK), lds, c->stream, c->sv, c->q, rc, c->stack_depth, c->d_cnt);
					else if (c->trav_mode == 1) hipLaunchKernelGGL(k_shadow<1>, dim3(grid), dim3(JP_BLOCK), lds, c->stream, c->sv, c->q, rc, c->stack_depth, c->d_cnt);
					else hipLaunchKernelGGL(k_shadow<0>, dim3(grid), dim3(JP_BLOCK), lds, c->stream, c->sv, c->q, rc, c->stack_depth, c->d_cnt);
					HIP_TRY(hipGetLastError());
				}
				cur ^= 1;
			}
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
		l->device = c->device; l->is_lane = true; l->n_cus = c->n_cus; l->blocks_per_cu = c->blocks_per_cu; l->opt = c->opt; l->opt_env = c->opt_env;
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
	l->have_scene = c->have_scene; l->sv = c->sv; l->stack_depth = c->stack_depth; l->stack_depth_q4 = c->stack_depth_q4; l->scene_in_lds = c->scene_in_lds; l->shade_prims_in_lds = c->shade_prims_in_lds;
	l->lds_bytes = c->lds_bytes; l->lds_bytes_shadow = c->lds_bytes_shadow; l->trav_mode = c->trav_mode; l->n_planes = c->n_planes;
	l->stack_lds_words = c->stack_lds_words;
	l->use_q4 = c->use_q4; l->q4_shadow = c->q4_shadow; l->cert = c->cert;
	l->persist = c->persist; l->vote = c->vote; l->shade_sort = c->shade_sort; l->class_mask = c->class_mask;
	l->has_null_material = c->has_null_material; l->tables_in_lds = c->tables_in_lds; l->stage_nee = c->stage_nee; l->shade_lds_bytes = c->shade_lds_bytes;
	l->profiling = c->profiling;
	l->opt = c->opt;                                                 // render_one(lane) reads max_slots / compact_regions from its own context
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
	if (c->opt.fused <= 0) return false;                             // JpOptions::fused
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
		if (c->opt.max_slots > 0) budget = std::min<size_t>(budget, (size_t)c->opt.max_slots * 16);
		const size_t PMAX = (size_t)1 << 26;
		const size_t pcap = std::min<size_t>(PMAX, std::max<size_t>((size_t)npix, budget / 16));
		int sbatch = (int)std::max<long long>(1, std::min<long long>(rp->spp, (long long)(pcap / (size_t)npix)));
		{ const int nb = (rp->spp + sbatch - 1) / sbatch; sbatch = (rp->spp + nb - 1) / nb; }       // equal batches
		// ---- job shape: R paths = PG pixels x S samples.  A wave's 64 lanes are 64 neighbouring pixels of one sample. ----
		unsigned int R = 1024;
		{ const int v = c->opt.fused_region; if (v >= JP_BLOCK && v <= 8192) R = (unsigned int)(v / JP_BLOCK) * JP_BLOCK; }
		int S = 16;
		{ const int v = c->opt.fused_job_spp; if (v >= 1 && v <= 128) S = v; }
		const int n_tab = 2 * c->sv.n_lights + 4 * c->sv.n_mats + (c->sv.n_mats + 3) / 4, n_tab_all = n_tab + (flat ? 8 * c->sv.n_prims : 0);
		const int deepE = (c->trav_mode != 5 && c->use_q4) ? c->stack_depth_q4 : c->stack_depth, deepS = (c->trav_mode != 5 && c->q4_shadow) ? c->stack_depth_q4 : (c->trav_mode == 3 ? (int)(c->lds_bytes_shadow / (JP_BLOCK * sizeof(int))) : c->stack_depth);
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
		{ const int v = c->opt.fused_workgroups; if (v >= 1 && v <= 16) per_cu = v; }
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
		rc.tiled = (rp->width % 16 == 0 && local_rows % 4 == 0 && PG % 64 == 0 && c->trav_mode != 2) ? 1 : 0;
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
	if (c->opt.lanes >= 1 && c->opt.lanes <= 4) forcedL = c->opt.lanes;
	if (c->opt.lane_rows >= 1 && c->opt.lane_rows <= 64) group = c->opt.lane_rows;
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
	out->certified_walk = (c->cert && c->persist != 0 && !c->cert_fell_back) ? 1 : 0;   // (what the refill kernels of the last render actually walked: the certified structures exist AND were used)
	out->certified_nodes = c->cert ? c->sv.n_q4 : 0; out->certified_eye_leaves = c->cert ? c->cert_eye_leaves : 0;
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
		// JpOptions::trace_walk (tests): 1 the binary tree, 2 the 8-wide tree, 3 the caller's tree verbatim; else what the render's closest-hit rays walk.  The one-ray-per-lane
		// kernels keep the whole stack in LDS: a 4-wide tree deeper than 64 KB of stack falls back to the binary / verbatim walk
		const size_t q4lds = (size_t)c->stack_depth_q4 * JP_BLOCK * sizeof(int); const int tw = c->opt.trace_walk;
		if (c->trav_mode == 3 && tw == 2) hipLaunchKernelGGL(k_trace<3>, dim3(grid), dim3(JP_BLOCK), c->lds_bytes_shadow, c->stream, c->sv, c->stack_depth, n, d_o, d_d, d_t0, d_t1, d_hit, d_t, d_prim, d_n);
		else if (c->trav_mode == 5 && c->cert && q4lds <= 64 * 1024 && tw != 3) hipLaunchKernelGGL(k_trace<6>, dim3(grid), dim3(JP_BLOCK), q4lds, c->stream, c->sv, c->stack_depth_q4, n, d_o, d_d, d_t0, d_t1, d_hit, d_t, d_prim, d_n);
		else if (c->trav_mode == 5) hipLaunchKernelGGL(k_trace<5>, dim3(grid), dim3(JP_BLOCK), c->lds_bytes, c->stream, c->sv, c->stack_depth, n, d_o, d_d, d_t0, d_t1, d_hit, d_t, d_prim, d_n);
		else if (c->use_q4 && q4lds <= 64 * 1024 && tw != 1) hipLaunchKernelGGL(k_trace<4>, dim3(grid), dim3(JP_BLOCK), q4lds, c->stream, c->sv, c->stack_depth_q4, n, d_o, d_d, d_t0, d_t1, d_hit, d_t, d_prim, d_n);
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

