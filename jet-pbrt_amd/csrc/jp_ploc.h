// jet-pbrt_amd/csrc/jp_ploc.h -- [round 3] device-side hierarchy build, second generation: PLOC (parallel locally-ordered clustering,
// Meister & Bittner 2018) instead of the Karras LBVH topology of jp_lbvh.h.  Replaces the host-side build of FScene::Preprocess
// (scene.cc:11-23 -> FBVH_NodeBase::Build, bvh.cc:23-84) for scenes handed over without a hierarchy.
//
// The LBVH splits a Morton range where the code's highest differing bit says; on the 280k-triangle scene that tree is 7-18 % slower to
// walk than the host's binned-SAH tree.  PLOC builds bottom-up by surface area: the clusters (at first the primitives in Morton order)
// each look for the neighbour within a window of +-kRadius positions whose union box has the least area; mutual nearest neighbours
// merge into a node; the array is compacted; repeat until one cluster is left.  Every step is a data-parallel kernel over the
// cluster array; the ids of the merged nodes come from a prefix sum, so the tree is a function of the input alone.
// Afterwards (all upward walks over parent links, no level-by-level passes):
//   k_ploc_first   position of every node's / primitive's subtree in depth-first order  = final device primitive order, in which the
//                  primitives of every subtree are contiguous -- what the leaf references (first << 4 | count - 1) need
//   k_ploc_emit    the 64-byte device nodes of jp_device.h (both children's padded boxes in the parent); subtrees of <= maxLeaf
//                  primitives become one leaf; node ids are reversed so that the root is node 0 and parents precede children
//   k_ploc_depth   height of the emitted tree (traversal stack)
// The 4-wide / 8-wide collapses of jp_lbvh.h then run on these nodes exactly as on the LBVH's.
#ifndef JP_PLOC_H
#define JP_PLOC_H

#define JP_PLOC_RADIUS 16

// ---- exclusive scan of a uint2 array (two counters per element), two levels: 1024 elements per workgroup -------------------------
__global__ void __launch_bounds__(256) k_scan2_block(const uint2* __restrict__ in, uint2* __restrict__ out, int n, uint2* __restrict__ tops)
{
	__shared__ unsigned int sx[256], sy[256];
	const int base = blockIdx.x * 1024 + threadIdx.x * 4;
	uint2 v[4]; unsigned int tx = 0, ty = 0;
	#pragma unroll
	for (int k = 0; k < 4; k++) { v[k] = base + k < n ? in[base + k] : make_uint2(0u, 0u); tx += v[k].x; ty += v[k].y; }
	sx[threadIdx.x] = tx; sy[threadIdx.x] = ty;
	__syncthreads();
	for (int off = 1; off < 256; off <<= 1)
	{
		const unsigned int ax = threadIdx.x >= (unsigned int)off ? sx[threadIdx.x - off] : 0u, ay = threadIdx.x >= (unsigned int)off ? sy[threadIdx.x - off] : 0u;
		__syncthreads();
		sx[threadIdx.x] += ax; sy[threadIdx.x] += ay;
		__syncthreads();
	}
	unsigned int ex = sx[threadIdx.x] - tx, ey = sy[threadIdx.x] - ty;
	#pragma unroll
	for (int k = 0; k < 4; k++) { if (base + k < n) out[base + k] = make_uint2(ex, ey); ex += v[k].x; ey += v[k].y; }
	if (threadIdx.x == 255) tops[blockIdx.x] = make_uint2(sx[255], sy[255]);
}
__global__ void __launch_bounds__(256) k_scan2_tops(uint2* tops, int nb, uint2* total)
{   // one workgroup: exclusive scan of the block totals in chunks of 256 with a running carry
	__shared__ unsigned int sx[256], sy[256];
	__shared__ unsigned int cx, cy;
	if (threadIdx.x == 0) { cx = 0; cy = 0; }
	__syncthreads();
	for (int base = 0; base < nb; base += 256)
	{
		const int i = base + threadIdx.x;
		const uint2 v = i < nb ? tops[i] : make_uint2(0u, 0u);
		sx[threadIdx.x] = v.x; sy[threadIdx.x] = v.y;
		__syncthreads();
		for (int off = 1; off < 256; off <<= 1)
		{
			const unsigned int ax = threadIdx.x >= (unsigned int)off ? sx[threadIdx.x - off] : 0u, ay = threadIdx.x >= (unsigned int)off ? sy[threadIdx.x - off] : 0u;
			__syncthreads();
			sx[threadIdx.x] += ax; sy[threadIdx.x] += ay;
			__syncthreads();
		}
		if (i < nb) tops[i] = make_uint2(cx + sx[threadIdx.x] - v.x, cy + sy[threadIdx.x] - v.y);
		__syncthreads();
		if (threadIdx.x == 255) { cx += sx[255]; cy += sy[255]; }
		__syncthreads();
	}
	if (threadIdx.x == 0) *total = make_uint2(cx, cy);
}
__global__ void __launch_bounds__(256) k_scan2_add(uint2* __restrict__ out, int n, const uint2* __restrict__ tops)
{
	const uint2 t = tops[blockIdx.x];
	const int base = blockIdx.x * 1024 + threadIdx.x * 4;
	#pragma unroll
	for (int k = 0; k < 4; k++) if (base + k < n) { uint2 v = out[base + k]; v.x += t.x; v.y += t.y; out[base + k] = v; }
}

// ---- clusters -------------------------------------------------------------------------------------------------------------------
// ref >= 0: interior node id (build numbering: order of creation); ref < 0: primitive at sorted position -ref - 1
struct PlocCluster { float lo[3]; int ref; float hi[3]; int cnt; };          // 32 bytes

__global__ void __launch_bounds__(256) k_ploc_init(const float4* __restrict__ lo, const float4* __restrict__ hi, int n, PlocCluster* __restrict__ c)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i >= n) return;
	const float4 l = lo[i], h = hi[i];
	PlocCluster k; k.lo[0] = l.x; k.lo[1] = l.y; k.lo[2] = l.z; k.hi[0] = h.x; k.hi[1] = h.y; k.hi[2] = h.z; k.ref = -i - 1; k.cnt = 1;
	c[i] = k;
}
__device__ __forceinline__ float ploc_union_area(const PlocCluster& a, const PlocCluster& b)
{
	const float dx = fmaxf(a.hi[0], b.hi[0]) - fminf(a.lo[0], b.lo[0]), dy = fmaxf(a.hi[1], b.hi[1]) - fminf(a.lo[1], b.lo[1]), dz = fmaxf(a.hi[2], b.hi[2]) - fminf(a.lo[2], b.lo[2]);
	return dx * dy + dy * dz + dz * dx;
}
// nearest neighbour within the window: least area of the union box; ties go to the partner i ^ 1, then to the nearer position, then to the
// smaller one -- a symmetric rule, so coincident primitives still pair up (0-1, 2-3, ...) instead of merging one pair per round
__global__ void __launch_bounds__(256) k_ploc_nn(const PlocCluster* __restrict__ c, int m, int radius, int* __restrict__ nn)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i >= m) return;
	const PlocCluster me = c[i];
	float bestA = 3.0e38f; int best = -1, bestRank = 0x7fffffff;
	const int j0 = max(0, i - radius), j1 = min(m - 1, i + radius);
	for (int j = j0; j <= j1; j++)
	{
		if (j == i) continue;
		const float A = ploc_union_area(me, c[j]);
		const int rank = j == (i ^ 1) ? 0 : 2 * abs(j - i) + (j > i ? 1 : 0);
		if (A < bestA || (A == bestA && rank < bestRank)) { bestA = A; best = j; bestRank = rank; }
	}
	nn[i] = best;
}
// flags: x = the cluster survives at its position (unchanged, or as the merged cluster of a pair), y = it is the lower half of a merging pair
__global__ void __launch_bounds__(256) k_ploc_flags(const int* __restrict__ nn, int m, uint2* __restrict__ flags)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i >= m) return;
	const int j = nn[i];
	const bool mutual = j >= 0 && nn[j] == i;
	flags[i] = make_uint2((!mutual || i < j) ? 1u : 0u, (mutual && i < j) ? 1u : 0u);
}
// node record of the build: children (refs in build numbering) and their boxes
struct PlocNode { float lo[2][3], hi[2][3]; int child[2]; int cnt; int parent; };     // 64 bytes
__global__ void __launch_bounds__(256) k_ploc_merge(const PlocCluster* __restrict__ c, const int* __restrict__ nn, const uint2* __restrict__ flags, const uint2* __restrict__ scan,
                                                    int m, int next_id, PlocCluster* __restrict__ out, PlocNode* __restrict__ nodes, int* __restrict__ parentLeaf)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i >= m) return;
	const uint2 f = flags[i];
	if (!f.x) return;                                            // the upper half of a merging pair: absorbed
	const uint2 at = scan[i];
	PlocCluster a = c[i];
	if (f.y)
	{
		const PlocCluster b = c[nn[i]];
		const int id = next_id + (int)at.y;
		PlocNode nd;
		for (int k = 0; k < 3; k++) { nd.lo[0][k] = a.lo[k]; nd.hi[0][k] = a.hi[k]; nd.lo[1][k] = b.lo[k]; nd.hi[1][k] = b.hi[k]; }
		nd.child[0] = a.ref; nd.child[1] = b.ref; nd.cnt = a.cnt + b.cnt; nd.parent = -1;
		nodes[id] = nd;
		if (a.ref >= 0) nodes[a.ref].parent = id; else parentLeaf[-a.ref - 1] = id;
		if (b.ref >= 0) nodes[b.ref].parent = id; else parentLeaf[-b.ref - 1] = id;
		for (int k = 0; k < 3; k++) { a.lo[k] = fminf(a.lo[k], b.lo[k]); a.hi[k] = fmaxf(a.hi[k], b.hi[k]); }
		a.ref = id; a.cnt = nd.cnt;
	}
	out[at.x] = a;
}

// depth-first position of the first primitive below every node and of every primitive: walking up, each time the walker is a RIGHT
// child the primitives of the left sibling come first
__global__ void __launch_bounds__(256) k_ploc_first(const PlocNode* __restrict__ nodes, const int* __restrict__ parentLeaf, int n, int* __restrict__ firstNode, int* __restrict__ posLeaf)
{
	const int t = blockIdx.x * 256 + threadIdx.x;
	if (t >= 2 * n - 1) return;
	const bool isLeaf = t < n;
	int ref = isLeaf ? -t - 1 : t - n;                            // the walker, as a child reference
	int p = isLeaf ? parentLeaf[t] : nodes[t - n].parent;
	int pos = 0;
	while (p >= 0)
	{
		const PlocNode* nd = &nodes[p];
		if (nd->child[1] == ref) { const int l = nd->child[0]; pos += l >= 0 ? nodes[l].cnt : 1; }
		ref = p; p = nd->parent;
	}
	if (isLeaf) posLeaf[t] = pos; else firstNode[t - n] = pos;
}
__global__ void __launch_bounds__(256) k_ploc_permute(const int* __restrict__ posLeaf, const int* __restrict__ sortedOrder, int n, const float4* __restrict__ primsS, const int4* __restrict__ metaS,
                                                      float4* __restrict__ prims, int4* __restrict__ meta, int* __restrict__ order)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i >= n) return;
	const int at = posLeaf[i];
	prims[4 * at] = primsS[4 * i]; prims[4 * at + 1] = primsS[4 * i + 1]; prims[4 * at + 2] = primsS[4 * i + 2]; prims[4 * at + 3] = primsS[4 * i + 3];
	meta[at] = metaS[i]; order[at] = sortedOrder[i];
}
// device node (jp_device.h) of every build node with more than maxLeaf primitives (and of the root): index = (n - 2) - build id
__global__ void __launch_bounds__(256) k_ploc_emit(const PlocNode* __restrict__ nodes, const int* __restrict__ firstNode, const int* __restrict__ posLeaf, int n, int maxLeaf, float4* __restrict__ out)
{
	const int id = blockIdx.x * 256 + threadIdx.x;
	if (id >= n - 1) return;
	const PlocNode nd = nodes[id];
	const int root = n - 2;
	if (id != root && nd.cnt <= maxLeaf) return;                  // inside a collapsed leaf
	float bl[2][3], bh[2][3]; int ref[2];
	for (int k = 0; k < 2; k++)
	{
		for (int a = 0; a < 3; a++) { bl[k][a] = nd.lo[k][a]; bh[k][a] = nd.hi[k][a]; lbvh_pad(bl[k][a], bh[k][a]); }
		const int c = nd.child[k];
		if (c < 0) ref[k] = -(((posLeaf[-c - 1] << 4) | 0) + 1);
		else { const int cnt = nodes[c].cnt; ref[k] = cnt <= maxLeaf ? -(((firstNode[c] << 4) | (cnt - 1)) + 1) : root - c; }
	}
	const int at = root - id;
	out[4 * at + 0] = make_float4(bl[0][0], bl[0][1], bl[0][2], bh[0][0]);
	out[4 * at + 1] = make_float4(bh[0][1], bh[0][2], bl[1][0], bl[1][1]);
	out[4 * at + 2] = make_float4(bl[1][2], bh[1][0], bh[1][1], bh[1][2]);
	out[4 * at + 3] = make_float4(__int_as_float(ref[0]), __int_as_float(ref[1]), 0, 0);
}
__global__ void __launch_bounds__(256) k_ploc_depth(const PlocNode* __restrict__ nodes, const int* __restrict__ parentLeaf, int n, int maxLeaf, int* height)
{
	const int j = blockIdx.x * 256 + threadIdx.x;
	int h = 0;
	if (j < n) for (int p = parentLeaf[j]; p >= 0; p = nodes[p].parent) if (p == n - 2 || nodes[p].cnt > maxLeaf) h++;
	for (int off = 32; off > 0; off >>= 1) h = max(h, __shfl_xor(h, off));
	if ((threadIdx.x & 63) == 0 && h > 0) atomicMax(height, h);
}

// Same contract as lbvh_build (jp_lbvh.h).  Returns hipErrorNotReady when the clustering does not finish within the round limit (the caller
// falls back to the LBVH topology); r is then untouched.
static hipError_t ploc_build(hipStream_t stream, const float4* prims0, const int4* meta0, int n, int maxLeaf, int opt_radius, int opt_max_rounds, LbvhResult& r, std::vector<int>& order)
{
	if (n < 2) return lbvh_build(stream, prims0, meta0, n, maxLeaf, r, order);
	hipError_t e = hipSuccess;
	std::vector<void*> tmp;
	auto dalloc = [&](void** p, size_t bytes, bool keep) -> bool { e = hipMalloc(p, std::max<size_t>(bytes, 16)); if (e != hipSuccess) return false; if (!keep) tmp.push_back(*p); return true; };
	auto cleanup = [&]() { for (void* p : tmp) hipFree(p); };
	auto bail = [&]() { cleanup(); if (r.d_nodes) hipFree(r.d_nodes); if (r.d_prims) hipFree(r.d_prims); if (r.d_meta) hipFree(r.d_meta); r = LbvhResult(); return e; };
	const size_t N = (size_t)n, NI = (size_t)(n - 1);
	float4 *lo0, *hi0, *lo, *hi, *primsS; int4* metaS; unsigned long long *keys, *keys2; int *vals, *vals2, *nn, *parentLeaf, *firstNode, *posLeaf, *height, *dorder;
	unsigned int* scene6; void* sorttmp = nullptr; PlocCluster *ca, *cb; PlocNode* nodes; uint2 *flags, *scan, *tops, *total;
	const size_t nblk = (N + 1023) / 1024;
	if (!dalloc((void**)&lo0, N * 16, false) || !dalloc((void**)&hi0, N * 16, false) || !dalloc((void**)&lo, N * 16, false) || !dalloc((void**)&hi, N * 16, false)
	    || !dalloc((void**)&primsS, N * 64, false) || !dalloc((void**)&metaS, N * 16, false) || !dalloc((void**)&keys, N * 8, false) || !dalloc((void**)&keys2, N * 8, false)
	    || !dalloc((void**)&vals, N * 4, false) || !dalloc((void**)&vals2, N * 4, false) || !dalloc((void**)&nn, N * 4, false) || !dalloc((void**)&parentLeaf, N * 4, false)
	    || !dalloc((void**)&firstNode, NI * 4, false) || !dalloc((void**)&posLeaf, N * 4, false) || !dalloc((void**)&height, 16, false) || !dalloc((void**)&dorder, N * 4, false)
	    || !dalloc((void**)&scene6, 32, false) || !dalloc((void**)&ca, N * sizeof(PlocCluster), false) || !dalloc((void**)&cb, N * sizeof(PlocCluster), false)
	    || !dalloc((void**)&nodes, NI * sizeof(PlocNode), false) || !dalloc((void**)&flags, N * 8, false) || !dalloc((void**)&scan, N * 8, false)
	    || !dalloc((void**)&tops, nblk * 8, false) || !dalloc((void**)&total, 16, false)
	    || !dalloc(&sorttmp, (size_t)256 * ((N + 256 * JP_RS_ITEMS - 1) / (256 * JP_RS_ITEMS)) * sizeof(unsigned int), false)
	    || !dalloc(&r.d_nodes, NI * 64, true) || !dalloc(&r.d_prims, N * 64, true) || !dalloc(&r.d_meta, N * 16, true))
		return bail();
	hipEvent_t e0, e1;
	if ((e = hipEventCreate(&e0)) != hipSuccess) return bail();
	if ((e = hipEventCreate(&e1)) != hipSuccess) { hipEventDestroy(e0); return bail(); }
	const int grid = (n + 255) / 256;
	hipEventRecord(e0, stream);
	hipLaunchKernelGGL(k_lbvh_init, dim3(1), dim3(64), 0, stream, scene6, height);
	hipMemsetAsync(r.d_nodes, 0, NI * 64, stream);
	hipLaunchKernelGGL(k_lbvh_bounds, dim3(grid), dim3(256), 0, stream, prims0, n, lo0, hi0, scene6);
	hipLaunchKernelGGL(k_lbvh_morton, dim3(grid), dim3(256), 0, stream, (const float4*)lo0, (const float4*)hi0, n, (const unsigned int*)scene6, keys, vals);
	lbvh_sort(stream, keys, keys2, vals, vals2, n, (unsigned int*)sorttmp);
	hipLaunchKernelGGL(k_lbvh_gather, dim3(grid), dim3(256), 0, stream, (const int*)vals2, n, prims0, meta0, (const float4*)lo0, (const float4*)hi0, primsS, metaS, lo, hi);
	hipLaunchKernelGGL(k_ploc_init, dim3(grid), dim3(256), 0, stream, (const float4*)lo, (const float4*)hi, n, ca);
	int radius = JP_PLOC_RADIUS;                                   // measured on the 280k-triangle scene (profiles/r03g_ploc_ab.txt)
	if (opt_radius >= 1 && opt_radius <= 256) radius = opt_radius;   // JpOptions::ploc_radius
	int max_rounds = 512;                                          // ~40 rounds for 280k primitives; the limit guards against a build that does not converge
	if (opt_max_rounds >= 1) max_rounds = opt_max_rounds;          // JpOptions::ploc_max_rounds (test hook: forces the LBVH fallback)
	int m = n, next_id = 0, rounds = 0;
	bool stuck = false;
	while (m > 1)
	{
		if (++rounds > max_rounds) { stuck = true; break; }
		const int g = (m + 255) / 256, nb = (m + 1023) / 1024;
		hipLaunchKernelGGL(k_ploc_nn, dim3(g), dim3(256), 0, stream, (const PlocCluster*)ca, m, radius, nn);
		hipLaunchKernelGGL(k_ploc_flags, dim3(g), dim3(256), 0, stream, (const int*)nn, m, flags);
		hipLaunchKernelGGL(k_scan2_block, dim3(nb), dim3(256), 0, stream, (const uint2*)flags, scan, m, tops);
		hipLaunchKernelGGL(k_scan2_tops, dim3(1), dim3(256), 0, stream, tops, nb, total);
		hipLaunchKernelGGL(k_scan2_add, dim3(nb), dim3(256), 0, stream, scan, m, (const uint2*)tops);
		hipLaunchKernelGGL(k_ploc_merge, dim3(g), dim3(256), 0, stream, (const PlocCluster*)ca, (const int*)nn, (const uint2*)flags, (const uint2*)scan, m, next_id, cb, nodes, parentLeaf);
		uint2 h_total;
		if ((e = hipMemcpyAsync(&h_total, total, 8, hipMemcpyDeviceToHost, stream)) != hipSuccess || (e = hipStreamSynchronize(stream)) != hipSuccess) { hipEventDestroy(e0); hipEventDestroy(e1); return bail(); }
		if (h_total.y == 0) { stuck = true; break; }             // no mutual pair: cannot happen for m >= 2 (the global minimum is mutual), guards a hang
		m = (int)h_total.x; next_id += (int)h_total.y;
		std::swap(ca, cb);
	}
	if (stuck || next_id != n - 1) { hipEventDestroy(e0); hipEventDestroy(e1); e = hipErrorNotReady; hipError_t keep = e; bail(); return keep; }
	hipLaunchKernelGGL(k_ploc_first, dim3((2 * n - 1 + 255) / 256), dim3(256), 0, stream, (const PlocNode*)nodes, (const int*)parentLeaf, n, firstNode, posLeaf);
	hipLaunchKernelGGL(k_ploc_permute, dim3(grid), dim3(256), 0, stream, (const int*)posLeaf, (const int*)vals2, n, (const float4*)primsS, (const int4*)metaS, (float4*)r.d_prims, (int4*)r.d_meta, dorder);
	hipLaunchKernelGGL(k_ploc_emit, dim3(grid), dim3(256), 0, stream, (const PlocNode*)nodes, (const int*)firstNode, (const int*)posLeaf, n, maxLeaf, (float4*)r.d_nodes);
	hipLaunchKernelGGL(k_ploc_depth, dim3(grid), dim3(256), 0, stream, (const PlocNode*)nodes, (const int*)parentLeaf, n, maxLeaf, height);
	hipEventRecord(e1, stream);
	order.resize(N);
	if ((e = hipMemcpyAsync(order.data(), dorder, N * 4, hipMemcpyDeviceToHost, stream)) != hipSuccess || (e = hipMemcpyAsync(&r.height, height, 4, hipMemcpyDeviceToHost, stream)) != hipSuccess)
	{ hipEventDestroy(e0); hipEventDestroy(e1); return bail(); }
	e = hipStreamSynchronize(stream);
	if (e == hipSuccess) e = hipGetLastError();
	if (e == hipSuccess) hipEventElapsedTime(&r.build_ms, e0, e1);
	hipEventDestroy(e0); hipEventDestroy(e1);
	if (e != hipSuccess) return bail();
	r.n_nodes = n - 1;
	cleanup();
	return hipSuccess;
}
#endif
