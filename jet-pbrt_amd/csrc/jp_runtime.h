// jet-pbrt_amd/csrc/jp_runtime.h -- host runtime, part 1 of 3: the context behind the C ABI (include/jetpbrt_amd.h), the options (JpOptions, ABI 7),
// the probes of the host libm the device reproduces, the gamma-threshold table, jp_create_context / jp_destroy_context / jp_set_options.
// Included by jp_kernels.hip (one translation unit: the kernels above, then this host code that launches them).
#pragma once

// =====================================================================================================================
// host side: context, scene upload, render loop, C ABI
// =====================================================================================================================
static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIP_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return fail(JP_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)); } while (0)

// tri-state switch of JpOptions: 0 default, 1 on, -1 off
static inline bool opt_flag(int v, bool dflt) { return v == 0 ? dflt : v > 0; }

// The environment as the initial value of the options (read ONCE per context, in jp_create_context): a test override, not an interface.
// kind 0: integer as it is; 1: switch ("0" = off -> -1, anything else on); 2: number where "0" means none (-> -1); 3: 64-bit integer;
// 4: the device tree by name ("lbvh" -> 2, anything else PLOC); 5: presence selects the value in `as`; 6: traversal mode (stored as 1 + mode)
static void options_from_environment(JpOptions& o)
{
	struct Item { const char* name; int kind; void* field; int as; };
	const Item items[] = {
		{ "JETPBRT_LANES", 0, &o.lanes, 0 }, { "JETPBRT_LANE_ROWS", 0, &o.lane_rows, 0 }, { "JETPBRT_BLOCKS_PER_CU", 0, &o.blocks_per_cu, 0 }, { "JETPBRT_MAX_SLOTS", 3, &o.max_slots, 0 },
		{ "JETPBRT_COMPACT_REGIONS", 1, &o.compact_regions, 0 }, { "JETPBRT_FUSED", 1, &o.fused, 0 }, { "JETPBRT_REGION", 0, &o.fused_region, 0 }, { "JETPBRT_JOB_SPP", 0, &o.fused_job_spp, 0 },
		{ "JETPBRT_FUSED_WGS", 0, &o.fused_workgroups, 0 }, { "JETPBRT_TRAVERSAL", 6, &o.traversal, 0 }, { "JETPBRT_Q4", 1, &o.q4, 0 }, { "JETPBRT_Q4_SHADOW", 1, &o.q4_shadow, 0 },
		{ "JETPBRT_PERSIST", 2, &o.persist, 0 }, { "JETPBRT_VOTE", 1, &o.vote, 0 }, { "JETPBRT_STACK_LDS", 0, &o.stack_lds_words, 0 }, { "JETPBRT_SHADE_SORT", 1, &o.shade_sort, 0 },
		{ "JETPBRT_DEVICE_TREE", 4, &o.device_tree, 0 }, { "JETPBRT_DEVICE_WIDE", 1, &o.device_wide, 0 }, { "JETPBRT_BVH_MAXLEAF", 0, &o.bvh_max_leaf, 0 },
		{ "JETPBRT_PLOC_RADIUS", 0, &o.ploc_radius, 0 }, { "JETPBRT_PLOC_MAX_ROUNDS", 0, &o.ploc_max_rounds, 0 }, { "JETPBRT_CERTIFIED", 1, &o.certified, 0 },
		{ "JETPBRT_CERT_SLACK", 7, &o.cert_slack, 0 }, { "JETPBRT_CERT_SLACK_EYE", 7, &o.cert_slack_eye, 0 }, { "JETPBRT_CERT_EYE", 7, &o.cert_eye_tau, 0 },
		{ "JETPBRT_SINCOSF", 2, &o.libm_sincosf, 0 }, { "JETPBRT_LIBM", 2, &o.libm_xbsdf, 0 },
		{ "JETPBRT_TRACE_BINARY", 5, &o.trace_walk, 1 }, { "JETPBRT_TRACE_WIDE", 5, &o.trace_walk, 2 }, { "JETPBRT_TRACE_VERBATIM", 5, &o.trace_walk, 3 }, { "JETPBRT_BOX_PAD", 7, &o.box_pad, 0 },
	};
	for (const Item& it : items)
	{
		const char* e = getenv(it.name);
		if (!e || !*e) continue;
		switch (it.kind)
		{
			case 0: *(int32_t*)it.field = atoi(e); break;
			case 1: *(int32_t*)it.field = atoi(e) != 0 ? 1 : -1; break;
			case 2: { const int v = atoi(e); *(int32_t*)it.field = v != 0 ? v : -1; break; }
			case 3: *(int64_t*)it.field = atoll(e); break;
			case 4: *(int32_t*)it.field = std::string(e) == "lbvh" ? 2 : 1; break;
			case 5: *(int32_t*)it.field = it.as; break;
			case 6: { const int v = atoi(e); if (v >= 0 && v <= 3) *(int32_t*)it.field = 1 + v; break; }
			case 7: { const float v = (float)atof(e); *(float*)it.field = v != 0.f ? v : -1.f; break; }
		}
	}
}

struct JpContext
{
	int device = 0;
	JpOptions opt{}, opt_env{};                                    // the options in force; their initial value (defaults + environment, jp_create_context)
	hipStream_t stream = nullptr;
	int n_cus = 256;
	// scene
	bool have_scene = false;
	SceneView sv; int stack_depth = 1, stack_depth_q4 = 0; bool scene_in_lds = false, shade_prims_in_lds = false; size_t lds_bytes = 0, lds_bytes_shadow = 0;
	void *d_flat = nullptr, *d_wide = nullptr, *d_q4 = nullptr; int trav_mode = 0;
	void* d_refbox = nullptr; bool cert_fell_back = false; bool cert = false; int cert_eye_leaves = 0;                                       // reference semantics, certified walk (Walker<6>): leaf boxes per primitive
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
	void** ps[] = { &c->d_flat, &c->d_wide, &c->d_q4, &c->d_refbox, &c->d_nodes, &c->d_prims, &c->d_meta, &c->d_mats, &c->d_mat_type, &c->d_lights, &c->d_shade_tab };
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
	std::memset(&c->opt_env, 0, sizeof(JpOptions)); c->opt_env.struct_bytes = (int32_t)sizeof(JpOptions);
	options_from_environment(c->opt_env);
	c->opt = c->opt_env;
	if (c->opt.blocks_per_cu >= 1 && c->opt.blocks_per_cu <= 256) { c->blocks_per_cu = c->opt.blocks_per_cu; c->bpc_from_env = true; }
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) c->n_cus = prop.multiProcessorCount;
	if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess
	    || hipMalloc((void**)&c->d_cnt, sizeof(DevCounters)) != hipSuccess)
	{ delete c; return fail(JP_ERR_DEVICE, "jp_create_context: stream/event/counter allocation failed"); }
	c->sincosf_mode = c->opt.libm_sincosf == 0 ? probe_host_sincosf() : (c->opt.libm_sincosf < 0 ? 0 : std::min(2, c->opt.libm_sincosf));
	{ hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(jp::g_sincosf_mode), &c->sincosf_mode, sizeof(int)); if (e != hipSuccess) { jp_destroy_context(c); return fail(JP_ERR_DEVICE, std::string("jp_create_context: hipMemcpyToSymbol: ") + hipGetErrorString(e)); } }
	c->libm_mode = c->opt.libm_xbsdf == 0 ? probe_host_libm() : (c->opt.libm_xbsdf < 0 ? 0 : (c->opt.libm_xbsdf & 3));
	{ hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(jp::xb::g_libm_mode), &c->libm_mode, sizeof(int)); if (e != hipSuccess) { jp_destroy_context(c); return fail(JP_ERR_DEVICE, std::string("jp_create_context: hipMemcpyToSymbol: ") + hipGetErrorString(e)); } }
	*out = c;
	return JP_OK;
}

// ABI 7: options by value.  Schedule fields take effect with the next jp_render*, traversal fields with the next jp_upload_scene; the libm fields
// re-select the device's transcription at once.  NULL restores the initial value (defaults + environment).
int jp_set_options(JpContext* c, const JpOptions* o)
{
	if (!c) return fail(JP_ERR_INVALID_ARGUMENT, "jp_set_options: null context");
	JpOptions n = c->opt_env;
	if (o)
	{
		if (o->struct_bytes < (int32_t)sizeof(int32_t) || o->struct_bytes > 4096) return fail(JP_ERR_INVALID_ARGUMENT, "jp_set_options: struct_bytes is not a struct size");
		std::memset(&n, 0, sizeof(n));
		std::memcpy(&n, o, std::min<size_t>(sizeof(n), (size_t)o->struct_bytes));
		n.struct_bytes = (int32_t)sizeof(JpOptions);
		if (n.lanes < 0 || n.lanes > 4 || n.lane_rows < 0 || n.lane_rows > 64 || n.blocks_per_cu < 0 || n.blocks_per_cu > 256 || n.max_slots < 0 || n.traversal < 0 || n.traversal > 4
		    || n.stack_lds_words < 0 || n.bvh_max_leaf < 0 || n.bvh_max_leaf > 16 || n.trace_walk < 0 || n.trace_walk > 3 || n.device_tree < 0 || n.device_tree > 2
		    || (n.persist > 0 && n.persist != 8 && n.persist != 16 && n.persist != 32))
			return fail(JP_ERR_INVALID_ARGUMENT, "jp_set_options: field out of range (see JpOptions in jetpbrt_amd.h)");
	}
	const bool libm_changed = n.libm_sincosf != c->opt.libm_sincosf || n.libm_xbsdf != c->opt.libm_xbsdf;
	c->opt = n;
	c->blocks_per_cu = (n.blocks_per_cu >= 1) ? n.blocks_per_cu : 16; c->bpc_from_env = n.blocks_per_cu >= 1;
	for (JpContext* l : c->lanes) { l->blocks_per_cu = c->blocks_per_cu; l->opt = c->opt; }
	if (libm_changed)
	{
		HIP_TRY(hipSetDevice(c->device));
		c->sincosf_mode = n.libm_sincosf == 0 ? probe_host_sincosf() : (n.libm_sincosf < 0 ? 0 : std::min(2, n.libm_sincosf));
		c->libm_mode = n.libm_xbsdf == 0 ? probe_host_libm() : (n.libm_xbsdf < 0 ? 0 : (n.libm_xbsdf & 3));
		HIP_TRY(hipStreamSynchronize(c->stream));
		HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(jp::g_sincosf_mode), &c->sincosf_mode, sizeof(int)));
		HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(jp::xb::g_libm_mode), &c->libm_mode, sizeof(int)));
	}
	return JP_OK;
}
int jp_get_options(JpContext* c, JpOptions* out)
{
	if (!c || !out) return fail(JP_ERR_INVALID_ARGUMENT, "jp_get_options: null argument");
	*out = c->opt; out->struct_bytes = (int32_t)sizeof(JpOptions);
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

