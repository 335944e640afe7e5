// jet-pbrt_amd/csrc/jp_lbvh.h -- device-side hierarchy build.
//
// Replaces the host-side tree build of FScene::Preprocess (scene.cc:11-23 -> FBVH_NodeBase::Build, bvh.cc:23-84) for
// scenes handed over WITHOUT a hierarchy (JpScene.n_bvh_nodes == 0): the primitive records are uploaded in creation
// order and the tree is built where it is used.  Closest-hit / occlusion results do not depend on the topology
// (SURVEY.md section 7), so the film is the same as with the host-built SAH tree up to the tie-break between
// coincident surfaces.
//
// Pipeline (all on the context stream, no host round trip until the final height read-back):
//   k_lbvh_bounds   exact extent of every primitive + scene extent (wave-reduced, ordered-uint atomics)
//   k_lbvh_morton   63-bit Morton code of the box centre (21 bits per axis)
//   lbvh_sort       own LSD radix sort of (code, primitive) pairs, 8 bits per pass, stable (k_rs_hist / k_rs_scan / k_rs_scatter)
//   k_lbvh_gather   primitive records / meta / boxes into sorted order (= device primitive order)
//   k_lbvh_hier     Karras 2012: one thread per interior node finds its range and split (duplicate codes are
//                   disambiguated by index, so the tree is a function of the sorted order alone)
//   k_lbvh_refit    bottom-up: the second thread to arrive at a node joins the child boxes and writes the device
//                   node record (children boxes in the parent, padded like the host path); subtrees of <= maxLeaf
//                   primitives collapse into one leaf (their primitives are contiguous in sorted order)
//   k_lbvh_depth    height of the emitted tree (sizes the per-lane LDS traversal stack)
#ifndef JP_LBVH_H
#define JP_LBVH_H

__device__ __forceinline__ unsigned int lbvh_ord(float f) { const unsigned int b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
__device__ __forceinline__ float lbvh_unord(unsigned int u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

__global__ void k_lbvh_init(unsigned int* scene6, int* height)
{
	if (threadIdx.x < 3) scene6[threadIdx.x] = 0xffffffffu; else if (threadIdx.x < 6) scene6[threadIdx.x] = 0u;
	if (threadIdx.x == 6) *height = 0;
}

__global__ void __launch_bounds__(256) k_lbvh_bounds(const float4* __restrict__ prims0, int n, float4* __restrict__ lo, float4* __restrict__ hi, unsigned int* scene6)
{
	const int p = blockIdx.x * 256 + threadIdx.x;
	float l[3] = { 3e38f, 3e38f, 3e38f }, h[3] = { -3e38f, -3e38f, -3e38f };
	if (p < n)
	{
		const float4 g0 = prims0[4 * p], g1 = prims0[4 * p + 1], g2 = prims0[4 * p + 2], g3 = prims0[4 * p + 3];
		const int type = __float_as_int(g3.w);
		if (type == JP_SHAPE_SPHERE || type == JP_SHAPE_DISK)          // disk record: (position, radius): its bounding sphere's box
		{
			l[0] = g0.x - g0.w; l[1] = g0.y - g0.w; l[2] = g0.z - g0.w; h[0] = g0.x + g0.w; h[1] = g0.y + g0.w; h[2] = g0.z + g0.w;
		}
		else
		{
			l[0] = fminf(fminf(g0.x, g1.x), g2.x); l[1] = fminf(fminf(g0.y, g1.y), g2.y); l[2] = fminf(fminf(g0.z, g1.z), g2.z);
			h[0] = fmaxf(fmaxf(g0.x, g1.x), g2.x); h[1] = fmaxf(fmaxf(g0.y, g1.y), g2.y); h[2] = fmaxf(fmaxf(g0.z, g1.z), g2.z);
			if (type == JP_SHAPE_RECTANGLE)
			{
				l[0] = fminf(l[0], g0.w); l[1] = fminf(l[1], g1.w); l[2] = fminf(l[2], g2.w);
				h[0] = fmaxf(h[0], g0.w); h[1] = fmaxf(h[1], g1.w); h[2] = fmaxf(h[2], g2.w);
			}
		}
		lo[p] = make_float4(l[0], l[1], l[2], 0); hi[p] = make_float4(h[0], h[1], h[2], 0);
	}
	#pragma unroll
	for (int a = 0; a < 3; a++)
	{
		float vl = l[a], vh = h[a];
		for (int off = 32; off > 0; off >>= 1) { vl = fminf(vl, __shfl_xor(vl, off)); vh = fmaxf(vh, __shfl_xor(vh, off)); }
		if ((threadIdx.x & 63) == 0 && vl <= vh) { atomicMin(&scene6[a], lbvh_ord(vl)); atomicMax(&scene6[3 + a], lbvh_ord(vh)); }
	}
}

__device__ __forceinline__ unsigned long long lbvh_spread21(unsigned int v)
{
	unsigned long long x = v & 0x1fffffu;
	x = (x | (x << 32)) & 0x1f00000000ffffull;
	x = (x | (x << 16)) & 0x1f0000ff0000ffull;
	x = (x | (x << 8)) & 0x100f00f00f00f00full;
	x = (x | (x << 4)) & 0x10c30c30c30c30c3ull;
	x = (x | (x << 2)) & 0x1249249249249249ull;
	return x;
}

__global__ void __launch_bounds__(256) k_lbvh_morton(const float4* __restrict__ lo, const float4* __restrict__ hi, int n, const unsigned int* __restrict__ scene6,
                                                     unsigned long long* __restrict__ keys, int* __restrict__ vals)
{
	const int p = blockIdx.x * 256 + threadIdx.x;
	if (p >= n) return;
	const float4 l = lo[p], h = hi[p];
	const float c[3] = { 0.5f * (l.x + h.x), 0.5f * (l.y + h.y), 0.5f * (l.z + h.z) };
	unsigned int q[3];
	#pragma unroll
	for (int a = 0; a < 3; a++)
	{
		const float sl = lbvh_unord(scene6[a]), sh = lbvh_unord(scene6[3 + a]);
		const float ext = sh - sl;
		float t = ext > 0.f ? (c[a] - sl) / ext : 0.f;
		t = fminf(fmaxf(t, 0.f), 1.f) * 2097152.f;
		q[a] = (unsigned int)fminf(t, 2097151.f);
	}
	keys[p] = (lbvh_spread21(q[0]) << 2) | (lbvh_spread21(q[1]) << 1) | lbvh_spread21(q[2]);
	vals[p] = p;
}

__global__ void __launch_bounds__(256) k_lbvh_gather(const int* __restrict__ order, int n, const float4* __restrict__ prims0, const int4* __restrict__ meta0,
                                                     const float4* __restrict__ lo0, const float4* __restrict__ hi0,
                                                     float4* __restrict__ prims, int4* __restrict__ meta, float4* __restrict__ lo, float4* __restrict__ hi)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i >= n) return;
	const int p = order[i];
	prims[4 * i] = prims0[4 * p]; prims[4 * i + 1] = prims0[4 * p + 1]; prims[4 * i + 2] = prims0[4 * p + 2]; prims[4 * i + 3] = prims0[4 * p + 3];
	meta[i] = meta0[p]; lo[i] = lo0[p]; hi[i] = hi0[p];
}

// length of the common prefix of the (code, index) keys of sorted positions i and j; -1 outside the array
__device__ __forceinline__ int lbvh_delta(const unsigned long long* __restrict__ keys, int n, int i, int j)
{
	if (j < 0 || j >= n) return -1;
	const unsigned long long x = keys[i] ^ keys[j];
	return x ? __clzll((long long)x) : 64 + __clz(i ^ j);
}

// child encoding in childL / childR: >= 0 interior node, < 0 leaf of sorted position -(c) - 1
__global__ void __launch_bounds__(256) k_lbvh_hier(const unsigned long long* __restrict__ keys, int n, int* __restrict__ childL, int* __restrict__ childR,
                                                   int* __restrict__ parentI, int* __restrict__ parentL, int* __restrict__ first, int* __restrict__ last)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i >= n - 1) return;
	const int d = lbvh_delta(keys, n, i, i + 1) - lbvh_delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
	const int dmin = lbvh_delta(keys, n, i, i - d);
	int lmax = 2;
	while (lbvh_delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
	int l = 0;
	for (int t = lmax >> 1; t >= 1; t >>= 1) if (lbvh_delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
	const int j = i + l * d;
	const int dnode = lbvh_delta(keys, n, i, j);
	int s = 0;
	for (int t = (l + 1) >> 1; ; t = (t + 1) >> 1)
	{
		if (lbvh_delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
		if (t == 1) break;
	}
	const int gamma = i + s * d + (d < 0 ? -1 : 0);
	const int f = i < j ? i : j, e = i < j ? j : i;
	first[i] = f; last[i] = e;
	if (gamma == f) { childL[i] = -gamma - 1; parentL[gamma] = i; } else { childL[i] = gamma; parentI[gamma] = i; }
	if (gamma + 1 == e) { childR[i] = -(gamma + 1) - 1; parentL[gamma + 1] = i; } else { childR[i] = gamma + 1; parentI[gamma + 1] = i; }
}

__device__ __forceinline__ void lbvh_store4(float4* p, float4 v)
{
	float* f = (float*)p;
	__hip_atomic_store(f + 0, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(f + 1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	__hip_atomic_store(f + 2, v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float4 lbvh_load4(const float4* p)
{
	float* f = (float*)p; float4 v;
	v.x = __hip_atomic_load(f + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); v.y = __hip_atomic_load(f + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	v.z = __hip_atomic_load(f + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); v.w = 0.f;
	return v;
}
// the relative pad of the host path (jp_upload_scene pad_box): >> ulp, keeps zero-extent boxes hittable
__device__ __forceinline__ void lbvh_pad(float& l, float& h)
{
	const float m = fmaxf(fabsf(l), fabsf(h)); const float e = m * 1e-6f + 1e-6f;
	l = l - e; h = h + e;
}

__global__ void __launch_bounds__(256) k_lbvh_refit(int n, int maxLeaf, const int* __restrict__ childL, const int* __restrict__ childR, const int* __restrict__ parentI,
                                                    const int* __restrict__ parentL, const int* __restrict__ first, const int* __restrict__ last,
                                                    const float4* __restrict__ leafLo, const float4* __restrict__ leafHi, float4* nodeLo, float4* nodeHi,
                                                    unsigned int* flag, float4* __restrict__ nodes)
{
	const int j = blockIdx.x * 256 + threadIdx.x;
	if (j >= n) return;
	int node = parentL[j];
	for (;;)
	{
		if (atomicAdd(&flag[node], 1u) == 0u) return;          // the first arrival leaves; the second finds both children complete
		__threadfence();
		float4 bl[2], bh[2]; int ref[2];
		#pragma unroll
		for (int k = 0; k < 2; k++)
		{
			const int c = k == 0 ? childL[node] : childR[node];
			if (c < 0) { const int leaf = -c - 1; bl[k] = leafLo[leaf]; bh[k] = leafHi[leaf]; ref[k] = -((leaf << 4) + 1); }
			else
			{
				bl[k] = lbvh_load4(&nodeLo[c]); bh[k] = lbvh_load4(&nodeHi[c]);
				const int cnt = last[c] - first[c] + 1;
				ref[k] = cnt <= maxLeaf ? -(((first[c] << 4) | (cnt - 1)) + 1) : c;
			}
		}
		lbvh_store4(&nodeLo[node], make_float4(fminf(bl[0].x, bl[1].x), fminf(bl[0].y, bl[1].y), fminf(bl[0].z, bl[1].z), 0));
		lbvh_store4(&nodeHi[node], make_float4(fmaxf(bh[0].x, bh[1].x), fmaxf(bh[0].y, bh[1].y), fmaxf(bh[0].z, bh[1].z), 0));
		if (node == 0 || last[node] - first[node] + 1 > maxLeaf)
		{
			lbvh_pad(bl[0].x, bh[0].x); lbvh_pad(bl[0].y, bh[0].y); lbvh_pad(bl[0].z, bh[0].z);
			lbvh_pad(bl[1].x, bh[1].x); lbvh_pad(bl[1].y, bh[1].y); lbvh_pad(bl[1].z, bh[1].z);
			nodes[4 * node + 0] = make_float4(bl[0].x, bl[0].y, bl[0].z, bh[0].x);
			nodes[4 * node + 1] = make_float4(bh[0].y, bh[0].z, bl[1].x, bl[1].y);
			nodes[4 * node + 2] = make_float4(bl[1].z, bh[1].x, bh[1].y, bh[1].z);
			nodes[4 * node + 3] = make_float4(__int_as_float(ref[0]), __int_as_float(ref[1]), 0, 0);
		}
		if (node == 0) return;
		__threadfence();
		node = parentI[node];
	}
}

// emitted interior nodes on the path root -> leaf (the traversal stack never holds more entries than that)
__global__ void __launch_bounds__(256) k_lbvh_depth(int n, int maxLeaf, const int* __restrict__ parentI, const int* __restrict__ parentL,
                                                    const int* __restrict__ first, const int* __restrict__ last, int* height)
{
	const int j = blockIdx.x * 256 + threadIdx.x;
	int h = 0;
	if (j < n)
		for (int node = parentL[j];; node = parentI[node])
		{
			if (node == 0 || last[node] - first[node] + 1 > maxLeaf) h++;
			if (node == 0) break;
		}
	for (int off = 32; off > 0; off >>= 1) h = max(h, __shfl_xor(h, off));
	if ((threadIdx.x & 63) == 0 && h > 0) atomicMax(height, h);
}

// a single primitive: synthetic interior root whose right child can never be hit (as the host path does)
__global__ void k_lbvh_single(const float4* __restrict__ lo, const float4* __restrict__ hi, float4* __restrict__ nodes)
{
	float4 l = lo[0], h = hi[0];
	lbvh_pad(l.x, h.x); lbvh_pad(l.y, h.y); lbvh_pad(l.z, h.z);
	const int ref = -1;                                        // leaf: first primitive 0, count 1
	nodes[0] = make_float4(l.x, l.y, l.z, h.x); nodes[1] = make_float4(h.y, h.z, 1e30f, 1e30f);
	nodes[2] = make_float4(1e30f, -1e30f, -1e30f, -1e30f); nodes[3] = make_float4(__int_as_float(ref), __int_as_float(ref), 0, 0);
}

struct LbvhResult { void* d_nodes = nullptr; void* d_prims = nullptr; void* d_meta = nullptr; int n_nodes = 0, height = 0; float build_ms = 0.f; };

// ---- LSD radix sort of (64-bit code, primitive index) pairs: 8 passes of 8 bits, stable --------------------------------------
// (round 1 called hipcub here; the sort is off the hot path -- 280k pairs once per scene upload -- but it is the step that
// decides the primitive order on the device, so it is own code like the rest.)  Per pass: k_rs_hist counts the digit values
// of every 2048-pair tile (LDS atomics), k_rs_scan turns the digit-major table of counts into global offsets, k_rs_scatter
// moves each tile's pairs, 256 at a time, to offset[digit][tile] + rank, the rank taken in tile order (wave match by eight
// ballots + per-wave counts through LDS), which keeps equal codes in index order: coincident primitives sort deterministically.
#define JP_RS_ITEMS 8
__global__ void __launch_bounds__(256) k_rs_hist(const unsigned long long* __restrict__ keys, int n, int shift, unsigned int* __restrict__ table, int ntiles)
{
	__shared__ unsigned int h[256];
	h[threadIdx.x] = 0;
	__syncthreads();
	const int base = blockIdx.x * 256 * JP_RS_ITEMS;
	for (int r = 0; r < JP_RS_ITEMS; r++)
	{
		const int i = base + r * 256 + threadIdx.x;
		if (i < n) atomicAdd(&h[(unsigned int)(keys[i] >> shift) & 255u], 1u);
	}
	__syncthreads();
	table[threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];       // digit-major: the exclusive scan gives [digit][tile] offsets
}
__global__ void __launch_bounds__(256) k_rs_scan(unsigned int* table, int total)
{   // one workgroup: exclusive scan of `total` counters in chunks of 256 with a running carry
	__shared__ unsigned int s[256];
	__shared__ unsigned int carry;
	if (threadIdx.x == 0) carry = 0;
	__syncthreads();
	for (int base = 0; base < total; base += 256)
	{
		const int i = base + threadIdx.x;
		const unsigned int v = i < total ? table[i] : 0u;
		s[threadIdx.x] = v;
		__syncthreads();
		for (int off = 1; off < 256; off <<= 1)
		{
			const unsigned int t = threadIdx.x >= (unsigned int)off ? s[threadIdx.x - off] : 0u;
			__syncthreads();
			s[threadIdx.x] += t;
			__syncthreads();
		}
		if (i < total) table[i] = carry + s[threadIdx.x] - v;
		__syncthreads();
		if (threadIdx.x == 255) carry += s[255];
		__syncthreads();
	}
}
__global__ void __launch_bounds__(256) k_rs_scatter(const unsigned long long* __restrict__ keys, const int* __restrict__ vals, int n, int shift,
                                                    const unsigned int* __restrict__ table, int ntiles, unsigned long long* __restrict__ keys_out, int* __restrict__ vals_out)
{
	__shared__ unsigned int s_base[256];                             // next free output position of each digit value for this tile
	__shared__ unsigned int s_cnt[4][256];                           // per wave: pairs of each digit value in the current round
	s_base[threadIdx.x] = table[threadIdx.x * ntiles + blockIdx.x];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const unsigned long long lt = (1ull << lane) - 1ull;
	const int base = blockIdx.x * 256 * JP_RS_ITEMS;
	for (int r = 0; r < JP_RS_ITEMS; r++)
	{
		for (int w = 0; w < 4; w++) s_cnt[w][threadIdx.x] = 0;
		__syncthreads();
		const int i = base + r * 256 + threadIdx.x;
		const bool valid = i < n;
		unsigned long long key = 0; int val = 0; unsigned int digit = 0;
		if (valid) { key = keys[i]; val = vals[i]; digit = (unsigned int)(key >> shift) & 255u; }
		unsigned long long m = __ballot(valid);                      // lanes of this wave with the same digit value
		for (int bit = 0; bit < 8; bit++) { const unsigned long long bm = __ballot(valid && ((digit >> bit) & 1u)); m &= ((digit >> bit) & 1u) ? bm : ~bm; }
		const unsigned int rank = (unsigned int)__popcll(m & lt);
		if (valid && rank == 0) s_cnt[wave][digit] = (unsigned int)__popcll(m);
		__syncthreads();
		if (valid)
		{
			unsigned int pos = s_base[digit] + rank;
			for (int w = 0; w < wave; w++) pos += s_cnt[w][digit];
			keys_out[pos] = key; vals_out[pos] = val;
		}
		__syncthreads();
		s_base[threadIdx.x] += s_cnt[0][threadIdx.x] + s_cnt[1][threadIdx.x] + s_cnt[2][threadIdx.x] + s_cnt[3][threadIdx.x];
		__syncthreads();
	}
}
// sorts (keys, vals) by key, bits [0, 64); the result is in (keys2, vals2); `table` holds 256 * ntiles counters
static void lbvh_sort(hipStream_t stream, unsigned long long* keys, unsigned long long* keys2, int* vals, int* vals2, int n, unsigned int* table)
{
	const int ntiles = (n + 256 * JP_RS_ITEMS - 1) / (256 * JP_RS_ITEMS);
	unsigned long long *ka = keys, *kb = keys2; int *va = vals, *vb = vals2;
	for (int pass = 0; pass < 8; pass++)
	{
		hipLaunchKernelGGL(k_rs_hist, dim3(ntiles), dim3(256), 0, stream, (const unsigned long long*)ka, n, 8 * pass, table, ntiles);
		hipLaunchKernelGGL(k_rs_scan, dim3(1), dim3(256), 0, stream, table, 256 * ntiles);
		hipLaunchKernelGGL(k_rs_scatter, dim3(ntiles), dim3(256), 0, stream, (const unsigned long long*)ka, (const int*)va, n, 8 * pass, (const unsigned int*)table, ntiles, kb, vb);
		std::swap(ka, kb); std::swap(va, vb);
	}
	// eight passes: the result is back in (keys, vals); the callers read (keys2, vals2)
	hipMemcpyAsync(keys2, keys, (size_t)n * 8, hipMemcpyDeviceToDevice, stream);
	hipMemcpyAsync(vals2, vals, (size_t)n * 4, hipMemcpyDeviceToDevice, stream);
}

// prims0 / meta0: device arrays in creation order (4 x float4 and one int4 per primitive).  On success the caller
// owns r.d_nodes / d_prims / d_meta and `order` holds, for each device (sorted) position, the creation-order index.
static hipError_t lbvh_build(hipStream_t stream, const float4* prims0, const int4* meta0, int n, int maxLeaf, LbvhResult& r, std::vector<int>& order)
{
	hipError_t e = hipSuccess;
	std::vector<void*> tmp;
	auto dalloc = [&](void** p, size_t bytes, bool keep) -> bool { e = hipMalloc(p, std::max<size_t>(bytes, 16)); if (e != hipSuccess) return false; if (!keep) tmp.push_back(*p); return true; };
	auto cleanup = [&]() { for (void* p : tmp) hipFree(p); };
	auto bail = [&]() { cleanup(); if (r.d_nodes) hipFree(r.d_nodes); if (r.d_prims) hipFree(r.d_prims); if (r.d_meta) hipFree(r.d_meta); r = LbvhResult(); return e; };
	float4 *lo0, *hi0, *lo, *hi, *nlo, *nhi; unsigned long long *keys, *keys2; int *vals, *vals2, *childL, *childR, *parentI, *parentL, *first, *last, *height;
	unsigned int *scene6, *flag; void* sorttmp = nullptr; size_t sortbytes = 0;
	const size_t N = (size_t)n, NI = (size_t)std::max(1, n - 1);
	if (!dalloc((void**)&lo0, N * 16, false) || !dalloc((void**)&hi0, N * 16, false) || !dalloc((void**)&lo, N * 16, false) || !dalloc((void**)&hi, N * 16, false)
	    || !dalloc((void**)&nlo, NI * 16, false) || !dalloc((void**)&nhi, NI * 16, false) || !dalloc((void**)&keys, N * 8, false) || !dalloc((void**)&keys2, N * 8, false)
	    || !dalloc((void**)&vals, N * 4, false) || !dalloc((void**)&vals2, N * 4, false) || !dalloc((void**)&childL, NI * 4, false) || !dalloc((void**)&childR, NI * 4, false)
	    || !dalloc((void**)&parentI, NI * 4, false) || !dalloc((void**)&parentL, N * 4, false) || !dalloc((void**)&first, NI * 4, false) || !dalloc((void**)&last, NI * 4, false)
	    || !dalloc((void**)&scene6, 32, false) || !dalloc((void**)&height, 16, false) || !dalloc((void**)&flag, NI * 4, false)
	    || !dalloc(&r.d_nodes, NI * 64, true) || !dalloc(&r.d_prims, N * 64, true) || !dalloc(&r.d_meta, N * 16, true))
		return bail();
	sortbytes = (size_t)256 * ((N + 256 * JP_RS_ITEMS - 1) / (256 * JP_RS_ITEMS)) * sizeof(unsigned int);
	if (!dalloc(&sorttmp, sortbytes, false)) return bail();
	hipEvent_t e0, e1;
	if ((e = hipEventCreate(&e0)) != hipSuccess) return bail();
	if ((e = hipEventCreate(&e1)) != hipSuccess) { hipEventDestroy(e0); return bail(); }
	const int grid = (n + 255) / 256;
	hipEventRecord(e0, stream);
	hipLaunchKernelGGL(k_lbvh_init, dim3(1), dim3(64), 0, stream, scene6, height);
	hipMemsetAsync(flag, 0, NI * 4, stream);
	hipMemsetAsync(r.d_nodes, 0, NI * 64, stream);
	hipLaunchKernelGGL(k_lbvh_bounds, dim3(grid), dim3(256), 0, stream, prims0, n, lo0, hi0, scene6);
	if (n == 1)
	{
		hipLaunchKernelGGL(k_lbvh_single, dim3(1), dim3(1), 0, stream, (const float4*)lo0, (const float4*)hi0, (float4*)r.d_nodes);
		hipMemcpyAsync(r.d_prims, prims0, 64, hipMemcpyDeviceToDevice, stream); hipMemcpyAsync(r.d_meta, meta0, 16, hipMemcpyDeviceToDevice, stream);
		order.assign(1, 0); r.n_nodes = 1; r.height = 1;
	}
	else
	{
		hipLaunchKernelGGL(k_lbvh_morton, dim3(grid), dim3(256), 0, stream, (const float4*)lo0, (const float4*)hi0, n, (const unsigned int*)scene6, keys, vals);
		lbvh_sort(stream, keys, keys2, vals, vals2, n, (unsigned int*)sorttmp);
		hipLaunchKernelGGL(k_lbvh_gather, dim3(grid), dim3(256), 0, stream, (const int*)vals2, n, prims0, meta0, (const float4*)lo0, (const float4*)hi0, (float4*)r.d_prims, (int4*)r.d_meta, lo, hi);
		hipLaunchKernelGGL(k_lbvh_hier, dim3(grid), dim3(256), 0, stream, (const unsigned long long*)keys2, n, childL, childR, parentI, parentL, first, last);
		hipLaunchKernelGGL(k_lbvh_refit, dim3(grid), dim3(256), 0, stream, n, maxLeaf, (const int*)childL, (const int*)childR, (const int*)parentI, (const int*)parentL,
		                   (const int*)first, (const int*)last, (const float4*)lo, (const float4*)hi, nlo, nhi, flag, (float4*)r.d_nodes);
		hipLaunchKernelGGL(k_lbvh_depth, dim3(grid), dim3(256), 0, stream, n, maxLeaf, (const int*)parentI, (const int*)parentL, (const int*)first, (const int*)last, height);
		r.n_nodes = n - 1;
	}
	hipEventRecord(e1, stream);
	if (n > 1)
	{
		order.resize(N);
		if ((e = hipMemcpyAsync(order.data(), vals2, N * 4, hipMemcpyDeviceToHost, stream)) != hipSuccess) { hipEventDestroy(e0); hipEventDestroy(e1); return bail(); }
		if ((e = hipMemcpyAsync(&r.height, height, 4, hipMemcpyDeviceToHost, stream)) != hipSuccess) { hipEventDestroy(e0); hipEventDestroy(e1); return bail(); }
	}
	e = hipStreamSynchronize(stream);
	if (e == hipSuccess) e = hipGetLastError();
	if (e == hipSuccess) hipEventElapsedTime(&r.build_ms, e0, e1);
	hipEventDestroy(e0); hipEventDestroy(e1);
	if (e != hipSuccess) return bail();
	cleanup();
	return hipSuccess;
}

// ---- 8-wide quantised tree for the any-hit (shadow) rays, built on the device from the binary tree above ------------------
// Same node format as the host builder of jp_upload_scene (traverse_wide, jp_device.h), same greedy collapse: a wide node
// starts from the two children of a binary node and keeps opening the inner child with the largest box while the result
// fits 8 slots; a leaf of n primitives takes ceil(n / 3) slots.  One thread builds one wide node; the tree grows level by
// level (breadth first), each level's threads allocating the wide indices of their inner children with one atomicAdd, so
// a node's inner children are contiguous and in slot order, as the traversal expects.  Slots are assigned in list order:
// the octant-ordered slots of the host builder only serve closest-hit ordering, which the shadow rays do not need.
// A leaf chunk whose primitives lie more than 24 records behind the node's primitive base (LBVH leaves of one wide node need
// not be neighbours in the sorted order) is wrapped in a wide node of its own.
struct WideItem { int bnode; unsigned int widx; };     // bnode >= 0: binary node; < 0: a leaf reference to wrap

__device__ __forceinline__ int wide_slots_of(int ref) { return ref >= 0 ? 1 : ((((-ref - 1) & 15) + 1) + 2) / 3; }
__device__ __forceinline__ float wide_area(const float* b) { const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2]; return dx * dy + dy * dz + dz * dx; }

__global__ void __launch_bounds__(64) k_wide_level(const float4* __restrict__ nodes, const WideItem* __restrict__ items, int n_items, uint32_t* __restrict__ wide,
                                                   unsigned int* wide_count, WideItem* __restrict__ next, unsigned int* next_count, unsigned int max_wide, int* fail)
{
	const int it = blockIdx.x * 64 + threadIdx.x;
	if (it >= n_items) return;
	const WideItem item = items[it];
	int ref[8]; float box[8][6]; int nch = 0;
	auto add_children_of = [&](int b) {
		const float4 n0 = nodes[4 * b], n1 = nodes[4 * b + 1], n2 = nodes[4 * b + 2], n3 = nodes[4 * b + 3];
		ref[nch] = __float_as_int(n3.x); box[nch][0] = n0.x; box[nch][1] = n0.y; box[nch][2] = n0.z; box[nch][3] = n0.w; box[nch][4] = n1.x; box[nch][5] = n1.y; nch++;
		ref[nch] = __float_as_int(n3.y); box[nch][0] = n1.z; box[nch][1] = n1.w; box[nch][2] = n2.x; box[nch][3] = n2.y; box[nch][4] = n2.z; box[nch][5] = n2.w; nch++;
	};
	float wrapbox[6];
	if (item.bnode >= 0)
	{
		add_children_of(item.bnode);
		for (;;)
		{
			int slots = 0; for (int k = 0; k < nch; k++) slots += wide_slots_of(ref[k]);
			int best = -1; float bestA = -1.f;
			for (int k = 0; k < nch; k++)
				if (ref[k] >= 0)
				{
					const float4 n3 = nodes[4 * ref[k] + 3];
					const int need = slots - 1 + wide_slots_of(__float_as_int(n3.x)) + wide_slots_of(__float_as_int(n3.y));
					const float A = wide_area(box[k]);
					if (need <= 8 && nch + 1 <= 8 && A > bestA) { bestA = A; best = k; }
				}
			if (best < 0) break;
			const int b = ref[best];
			for (int k = best; k + 1 < nch; k++) { ref[k] = ref[k + 1]; for (int a = 0; a < 6; a++) box[k][a] = box[k + 1][a]; }
			nch--;
			add_children_of(b);
		}
	}
	else
	{   // wrapper: the box comes with the item through the `next` array's companion (stored in the wide node slot itself beforehand)
		ref[0] = item.bnode; nch = 1;
		const float* wb = (const float*)&wide[(size_t)item.widx * 20];     // the parent parked the leaf's box here
		for (int a = 0; a < 6; a++) { box[0][a] = wb[a]; wrapbox[a] = wb[a]; }
	}
	// primitive base: the smallest first primitive among the direct leaf children; leaves too far behind it are wrapped
	int prim_base = 0x7fffffff;
	for (int k = 0; k < nch; k++) if (ref[k] < 0) { const int first = (-ref[k] - 1) >> 4; if (first < prim_base) prim_base = first; }
	bool wrap[8];
	for (int k = 0; k < nch; k++)
	{
		wrap[k] = false;
		if (ref[k] < 0 && item.bnode >= 0) { const int e = -ref[k] - 1, first = e >> 4, cnt = (e & 15) + 1; if (first - prim_base + cnt > 24) wrap[k] = true; }
	}
	// node box and scales
	float lo[3] = { 1e30f, 1e30f, 1e30f }, hi[3] = { -1e30f, -1e30f, -1e30f };
	for (int k = 0; k < nch; k++) for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], box[k][a]); hi[a] = fmaxf(hi[a], box[k][3 + a]); }
	int eb[3]; float sc[3];
	for (int a = 0; a < 3; a++)
	{
		int e = (int)ceilf(log2f(fmaxf((hi[a] - lo[a]) / 255.f, 1e-30f)));
		// log2f is approximate: make sure 255 steps really span the extent
		e = max(-120, min(120, e));
		while (e < 120 && fmaf(255.f, ldexpf(1.0f, e), lo[a]) < hi[a]) e++;
		eb[a] = e + 127; sc[a] = ldexpf(1.0f, e);
	}
	// slots in list order: inner children (and wrapped leaves) first count, then leaf chunks
	unsigned int ninner = 0; for (int k = 0; k < nch; k++) if (ref[k] >= 0 || wrap[k]) ninner++;
	unsigned int child_base = 0;
	if (ninner) { child_base = atomicAdd(wide_count, ninner); if (child_base + ninner > max_wide) { *fail = 1; return; } }
	unsigned int nbase = ninner ? atomicAdd(next_count, ninner) : 0u;
	unsigned char metaB[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, ql[3][8], qh[3][8];
	for (int sl = 0; sl < 8; sl++) for (int a = 0; a < 3; a++) { ql[a][sl] = 255; qh[a][sl] = 0; }
	unsigned int imask = 0, rank = 0; int sl = 0;
	auto quantise = [&](int slot, const float* b) {
		for (int a = 0; a < 3; a++)
		{
			int q0 = (int)floorf((b[a] - lo[a]) / sc[a]), q1 = (int)ceilf((b[3 + a] - lo[a]) / sc[a]);
			q0 = max(0, min(255, q0)); q1 = max(0, min(255, q1));
			while (q0 > 0 && fmaf((float)q0, sc[a], lo[a]) > b[a]) q0--;
			while (q1 < 255 && fmaf((float)q1, sc[a], lo[a]) < b[3 + a]) q1++;
			if (fmaf((float)q1, sc[a], lo[a]) < b[3 + a]) *fail = 1;
			ql[a][slot] = (unsigned char)q0; qh[a][slot] = (unsigned char)q1;
		}
	};
	for (int k = 0; k < nch; k++)
	{
		if (ref[k] >= 0 || wrap[k])
		{
			if (sl >= 8) { *fail = 1; return; }
			imask |= 1u << sl; metaB[sl] = (unsigned char)(0x20 | (24 + sl));
			WideItem ni; ni.bnode = ref[k]; ni.widx = child_base + rank;
			next[nbase + rank] = ni;
			if (wrap[k]) { float* wb = (float*)&wide[(size_t)ni.widx * 20]; for (int a = 0; a < 6; a++) wb[a] = box[k][a]; }   // park the leaf's box for the wrapper
			rank++;
			quantise(sl, box[k]); sl++;
		}
		else
		{
			const int e = -ref[k] - 1, first = e >> 4, cnt = (e & 15) + 1;
			for (int c0 = 0; c0 < cnt; c0 += 3)
			{
				if (sl >= 8) { *fail = 1; return; }
				const int cc = min(3, cnt - c0), off = first + c0 - prim_base;
				if (off < 0 || off + cc > 24) { *fail = 1; return; }
				metaB[sl] = (unsigned char)((((1u << cc) - 1u) << 5) | (unsigned int)off);
				quantise(sl, box[k]); sl++;
			}
		}
	}
	(void)wrapbox;
	uint32_t* w = &wide[(size_t)item.widx * 20];
	auto pack4 = [](const unsigned char* v) { return (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24); };
	w[0] = __float_as_uint(lo[0]); w[1] = __float_as_uint(lo[1]); w[2] = __float_as_uint(lo[2]);
	w[3] = (uint32_t)eb[0] | ((uint32_t)eb[1] << 8) | ((uint32_t)eb[2] << 16) | (imask << 24);
	w[4] = child_base; w[5] = prim_base == 0x7fffffff ? 0u : (uint32_t)prim_base; w[6] = pack4(metaB); w[7] = pack4(metaB + 4);
	w[8] = pack4(ql[0]); w[9] = pack4(ql[0] + 4); w[10] = pack4(ql[1]); w[11] = pack4(ql[1] + 4);
	w[12] = pack4(ql[2]); w[13] = pack4(ql[2] + 4); w[14] = pack4(qh[0]); w[15] = pack4(qh[0] + 4);
	w[16] = pack4(qh[1]); w[17] = pack4(qh[1] + 4); w[18] = pack4(qh[2]); w[19] = pack4(qh[2] + 4);
}

struct WideResult { void* d_wide = nullptr; int n_wide = 0, height = 0; float build_ms = 0.f; };

// nodes: the device binary tree of lbvh_build (root = node 0, which is always interior there).  On failure (a box that cannot
// be quantised conservatively, index overflow) r.d_wide stays null and the caller keeps the binary tree for the shadow rays.
static hipError_t lbvh_build_wide(hipStream_t stream, const float4* nodes, int n_prims, WideResult& r)
{
	r = WideResult();
	if (n_prims < 2) return hipSuccess;
	hipError_t e;
	const unsigned int max_wide = (unsigned int)std::max(16, 2 * n_prims);
	uint32_t* wide = nullptr; WideItem *fa = nullptr, *fb = nullptr; unsigned int* ctr = nullptr; int* fail = nullptr;
	auto bail = [&](hipError_t err) { if (wide) hipFree(wide); if (fa) hipFree(fa); if (fb) hipFree(fb); if (ctr) hipFree(ctr); if (fail) hipFree(fail); return err; };
	if ((e = hipMalloc((void**)&wide, (size_t)max_wide * 80)) != hipSuccess || (e = hipMalloc((void**)&fa, (size_t)max_wide * sizeof(WideItem))) != hipSuccess
	    || (e = hipMalloc((void**)&fb, (size_t)max_wide * sizeof(WideItem))) != hipSuccess || (e = hipMalloc((void**)&ctr, 16)) != hipSuccess || (e = hipMalloc((void**)&fail, 16)) != hipSuccess)
		return bail(e);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipEventRecord(e0, stream);
	hipMemsetAsync(wide, 0, (size_t)max_wide * 80, stream);
	hipMemsetAsync(fail, 0, 16, stream);
	const WideItem rootItem = { 0, 0u };
	unsigned int h_ctr[2] = { 1u, 0u };                            // wide_count (root allocated), next_count
	hipMemcpyAsync(fa, &rootItem, sizeof(rootItem), hipMemcpyHostToDevice, stream);
	int n_items = 1, levels = 0; int h_fail = 0;
	WideItem *cur = fa, *nxt = fb;
	while (n_items > 0 && levels < 64)
	{
		hipMemcpyAsync(ctr, h_ctr, 8, hipMemcpyHostToDevice, stream);
		hipLaunchKernelGGL(k_wide_level, dim3((n_items + 63) / 64), dim3(64), 0, stream, nodes, (const WideItem*)cur, n_items, wide, ctr, nxt, ctr + 1, max_wide, fail);
		hipMemcpyAsync(h_ctr, ctr, 8, hipMemcpyDeviceToHost, stream);
		hipMemcpyAsync(&h_fail, fail, 4, hipMemcpyDeviceToHost, stream);
		if ((e = hipStreamSynchronize(stream)) != hipSuccess) { hipEventDestroy(e0); hipEventDestroy(e1); return bail(e); }
		levels++;
		if (h_fail) break;
		n_items = (int)h_ctr[1]; h_ctr[1] = 0u;
		std::swap(cur, nxt);
	}
	hipEventRecord(e1, stream); hipStreamSynchronize(stream);
	hipEventElapsedTime(&r.build_ms, e0, e1); hipEventDestroy(e0); hipEventDestroy(e1);
	hipFree(fa); hipFree(fb); hipFree(ctr); hipFree(fail);
	if (h_fail || levels >= 64) { hipFree(wide); return hipSuccess; }
	r.d_wide = wide; r.n_wide = (int)h_ctr[0]; r.height = levels;
	return hipSuccess;
}
// ---- [round 3] 4-wide quantised tree (Walker<4>, jp_device.h) collapsed on the device from the binary tree above --------------------
// The same node format and the same collapse as the host builder of jp_upload_scene: from binary node b its two children, then the
// interior child with the largest box is opened while fewer than four slots are taken; every child box gets 1e-6 of the node's extent
// before it is quantised outward.  One thread builds one node, the tree grows level by level, a level's threads allocate the indices of
// their interior children with one atomicAdd.  Leaves keep the binary tree's references (first << 4 | count - 1).
__global__ void __launch_bounds__(64) k_q4_level(const float4* __restrict__ nodes, const WideItem* __restrict__ items, int n_items, uint32_t* __restrict__ q4,
                                                 unsigned int* q4_count, WideItem* __restrict__ next, unsigned int* next_count, unsigned int max_nodes, int* fail)
{
	const int it = blockIdx.x * 64 + threadIdx.x;
	if (it >= n_items) return;
	const WideItem item = items[it];
	int ref[4]; float box[4][6]; int nch = 0;
	auto add_children_of = [&](int b, int at0, int at1) {
		const float4 n0 = nodes[4 * b], n1 = nodes[4 * b + 1], n2 = nodes[4 * b + 2], n3 = nodes[4 * b + 3];
		ref[at0] = __float_as_int(n3.x); box[at0][0] = n0.x; box[at0][1] = n0.y; box[at0][2] = n0.z; box[at0][3] = n0.w; box[at0][4] = n1.x; box[at0][5] = n1.y;
		ref[at1] = __float_as_int(n3.y); box[at1][0] = n1.z; box[at1][1] = n1.w; box[at1][2] = n2.x; box[at1][3] = n2.y; box[at1][4] = n2.z; box[at1][5] = n2.w;
	};
	add_children_of(item.bnode, 0, 1); nch = 2;
	while (nch < 4)
	{
		int best = -1; float bestA = -1.f;
		for (int k = 0; k < nch; k++) if (ref[k] >= 0) { const float A = wide_area(box[k]); if (A > bestA) { bestA = A; best = k; } }
		if (best < 0) break;
		add_children_of(ref[best], best, nch); nch++;
	}
	float lo[3] = { 1e30f, 1e30f, 1e30f }, hi[3] = { -1e30f, -1e30f, -1e30f };
	for (int k = 0; k < nch; k++) for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], box[k][a]); hi[a] = fmaxf(hi[a], box[k][3 + a]); }
	for (int k = 0; k < nch; k++) for (int a = 0; a < 3; a++) { const float ex = 1e-6f * (hi[a] - lo[a]); box[k][a] -= ex; box[k][3 + a] += ex; }
	for (int a = 0; a < 3; a++) { const float ex = 1e-6f * (hi[a] - lo[a]); lo[a] -= ex; hi[a] += ex; }
	int eb[3]; float sc[3];
	for (int a = 0; a < 3; a++)
	{
		int e = (int)ceilf(log2f(fmaxf((hi[a] - lo[a]) / 255.f, 1e-30f)));
		e = max(-120, min(120, e));
		while (e < 120 && fmaf(255.f, ldexpf(1.0f, e), lo[a]) < hi[a]) e++;      // log2f is approximate: 255 steps must span the extent
		eb[a] = e + 127; sc[a] = ldexpf(1.0f, e);
	}
	unsigned int ninner = 0; for (int k = 0; k < nch; k++) if (ref[k] >= 0) ninner++;
	unsigned int child_base = 0, nbase = 0;
	if (ninner)
	{
		child_base = atomicAdd(q4_count, ninner); if (child_base + ninner > max_nodes) { *fail = 1; return; }
		nbase = atomicAdd(next_count, ninner);
	}
	unsigned char ql[3][4], qh[3][4]; uint32_t refs[4] = { 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu }, valid = 0, rank = 0;   // unused slots: the empty box (255 > 0) and primitive 0 as a one-primitive leaf (-1): WalkerQ4 tests no valid bit
	for (int k = 0; k < 4; k++) for (int a = 0; a < 3; a++) { ql[a][k] = 255; qh[a][k] = 0; }
	for (int k = 0; k < nch; k++)
	{
		valid |= 1u << k;
		if (ref[k] >= 0) { WideItem ni; ni.bnode = ref[k]; ni.widx = child_base + rank; next[nbase + rank] = ni; refs[k] = ni.widx; rank++; }
		else refs[k] = (uint32_t)ref[k];
		for (int a = 0; a < 3; a++)
		{
			int q0 = (int)floorf((box[k][a] - lo[a]) / sc[a]), q1 = (int)ceilf((box[k][3 + a] - lo[a]) / sc[a]);
			q0 = max(0, min(255, q0)); q1 = max(0, min(255, q1));
			while (q0 > 0 && fmaf((float)q0, sc[a], lo[a]) > box[k][a]) q0--;
			while (q1 < 255 && fmaf((float)q1, sc[a], lo[a]) < box[k][3 + a]) q1++;
			if (fmaf((float)q1, sc[a], lo[a]) < box[k][3 + a] || fmaf((float)q0, sc[a], lo[a]) > box[k][a]) *fail = 1;
			ql[a][k] = (unsigned char)q0; qh[a][k] = (unsigned char)q1;
		}
	}
	auto pack4 = [](const unsigned char* v) { return (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24); };
	uint32_t* w = &q4[(size_t)item.widx * 16];
	w[0] = __float_as_uint(lo[0]); w[1] = __float_as_uint(lo[1]); w[2] = __float_as_uint(lo[2]);
	w[3] = (uint32_t)eb[0] | ((uint32_t)eb[1] << 8) | ((uint32_t)eb[2] << 16) | (valid << 24);
	w[4] = refs[0]; w[5] = refs[1]; w[6] = refs[2]; w[7] = refs[3];
	w[8] = pack4(ql[0]); w[9] = pack4(ql[1]); w[10] = pack4(ql[2]); w[11] = pack4(qh[0]);
	w[12] = pack4(qh[1]); w[13] = pack4(qh[2]); w[14] = 0; w[15] = 0;
}

// nodes: the device binary tree (root = node 0, always interior).  On failure r.d_wide stays null and the caller keeps the binary tree.
static hipError_t lbvh_build_q4(hipStream_t stream, const float4* nodes, int n_prims, WideResult& r)
{
	r = WideResult();
	if (n_prims < 2) return hipSuccess;
	hipError_t e;
	const unsigned int max_nodes = (unsigned int)std::max(16, n_prims);
	uint32_t* q4 = nullptr; WideItem *fa = nullptr, *fb = nullptr; unsigned int* ctr = nullptr; int* fail = nullptr;
	auto bail = [&](hipError_t err) { if (q4) hipFree(q4); if (fa) hipFree(fa); if (fb) hipFree(fb); if (ctr) hipFree(ctr); if (fail) hipFree(fail); return err; };
	if ((e = hipMalloc((void**)&q4, (size_t)max_nodes * 64)) != hipSuccess || (e = hipMalloc((void**)&fa, (size_t)max_nodes * sizeof(WideItem))) != hipSuccess
	    || (e = hipMalloc((void**)&fb, (size_t)max_nodes * sizeof(WideItem))) != hipSuccess || (e = hipMalloc((void**)&ctr, 16)) != hipSuccess || (e = hipMalloc((void**)&fail, 16)) != hipSuccess)
		return bail(e);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipEventRecord(e0, stream);
	hipMemsetAsync(q4, 0, (size_t)max_nodes * 64, stream);
	hipMemsetAsync(fail, 0, 16, stream);
	const WideItem rootItem = { 0, 0u };
	unsigned int h_ctr[2] = { 1u, 0u };
	hipMemcpyAsync(fa, &rootItem, sizeof(rootItem), hipMemcpyHostToDevice, stream);
	int n_items = 1, levels = 0; int h_fail = 0;
	WideItem *cur = fa, *nxt = fb;
	while (n_items > 0 && levels < 64)
	{
		hipMemcpyAsync(ctr, h_ctr, 8, hipMemcpyHostToDevice, stream);
		hipLaunchKernelGGL(k_q4_level, dim3((n_items + 63) / 64), dim3(64), 0, stream, nodes, (const WideItem*)cur, n_items, q4, ctr, nxt, ctr + 1, max_nodes, fail);
		hipMemcpyAsync(h_ctr, ctr, 8, hipMemcpyDeviceToHost, stream);
		hipMemcpyAsync(&h_fail, fail, 4, hipMemcpyDeviceToHost, stream);
		if ((e = hipStreamSynchronize(stream)) != hipSuccess) { hipEventDestroy(e0); hipEventDestroy(e1); return bail(e); }
		levels++;
		if (h_fail) break;
		n_items = (int)h_ctr[1]; h_ctr[1] = 0u;
		std::swap(cur, nxt);
	}
	hipEventRecord(e1, stream); hipStreamSynchronize(stream);
	hipEventElapsedTime(&r.build_ms, e0, e1); hipEventDestroy(e0); hipEventDestroy(e1);
	hipFree(fa); hipFree(fb); hipFree(ctr); hipFree(fail);
	if (h_fail || levels >= 64) { hipFree(q4); return hipSuccess; }
	r.d_wide = q4; r.n_wide = (int)h_ctr[0]; r.height = levels;
	return hipSuccess;
}
#endif
