// jet-pbrt_amd/host/bvh_build.cc -- binned-SAH BVH builder producing the flat node arrays of JpScene.
//
// Replaces the reference's build (bvh.h:54-91: random split axis from rand(), std::sort, median split,
// leaves <= 5) with a surface-area-heuristic build: the device traversal is ordered with early-out, so tree
// quality matters, and closest-hit / occlusion RESULTS do not depend on topology.
// Nodes are emitted in depth-first order (left child = parent + 1) for locality.
#include "jetpbrt.h"

#include <algorithm>
#include <cstring>
#include <future>

namespace jetpbrt
{
namespace
{
struct B { float mn[3], mx[3]; };
inline B emptyB() { B b; for (int a = 0; a < 3; a++) { b.mn[a] = std::numeric_limits<float>::max(); b.mx[a] = std::numeric_limits<float>::lowest(); } return b; }
inline void grow(B& b, const B& o) { for (int a = 0; a < 3; a++) { b.mn[a] = std::min(b.mn[a], o.mn[a]); b.mx[a] = std::max(b.mx[a], o.mx[a]); } }
inline void growP(B& b, const float* p) { for (int a = 0; a < 3; a++) { b.mn[a] = std::min(b.mn[a], p[a]); b.mx[a] = std::max(b.mx[a], p[a]); } }
inline float area(const B& b)
{
	float dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
	if (dx < 0 || dy < 0 || dz < 0) return 0.f;
	return 2.f * (dx * dy + dy * dz + dz * dx);
}

// Subtrees over disjoint index ranges are independent, so the top levels fork: each side of a split is built by its own
// Builder into a private FlatBVH (sharing the read-only boxes / centroids and the index array, whose ranges are disjoint) and
// the two results are appended left first -- the node and leaf order of the sequential depth-first build, whatever the thread
// timing, so the tree does not depend on the number of threads.
struct Builder
{
	const std::vector<B>& pb; std::vector<float>& cen; std::vector<int32_t>& idx; FlatBVH& out; int maxLeaf;
	float travCost = 1.0f;                                       // cost of a node visit in primitive tests (JETPBRT_SAH_TRAV_COST)
	Builder(const std::vector<B>& pb, std::vector<float>& cen, std::vector<int32_t>& idx, FlatBVH& out, int maxLeaf) : pb(pb), cen(cen), idx(idx), out(out), maxLeaf(maxLeaf) {}
	int append(const FlatBVH& sub)                                 // returns the index the sub-tree's root gets
	{
		const int off = (int)out.left.size(), poff = (int)out.prim_index.size();
		out.bounds.insert(out.bounds.end(), sub.bounds.begin(), sub.bounds.end());
		for (size_t i = 0; i < sub.left.size(); i++)
		{
			if (sub.left[i] < 0) { out.left.push_back(-((-sub.left[i] - 1) + poff) - 1); out.right.push_back(sub.right[i]); }
			else { out.left.push_back(sub.left[i] + off); out.right.push_back(sub.right[i] + off); }
		}
		out.prim_index.insert(out.prim_index.end(), sub.prim_index.begin(), sub.prim_index.end());
		return off;
	}
	int emit(const B& b)
	{
		int n = (int)out.left.size();
		out.left.push_back(0); out.right.push_back(0);
		for (int a = 0; a < 3; a++) out.bounds.push_back(b.mn[a]);
		for (int a = 0; a < 3; a++) out.bounds.push_back(b.mx[a]);
		return n;
	}
	void leaf(int node, int start, int end)
	{
		int first = (int)out.prim_index.size();
		for (int i = start; i < end; i++) out.prim_index.push_back(idx[i]);
		out.left[node] = -first - 1; out.right[node] = end - start;
	}
	int build(int start, int end, int fork = 0)
	{
		B nb = emptyB(), cb = emptyB();
		for (int i = start; i < end; i++) { grow(nb, pb[idx[i]]); growP(cb, &cen[3 * idx[i]]); }
		int node = emit(nb);
		int n = end - start;
		if (n <= 1) { leaf(node, start, end); return node; }

		// binned SAH over the centroid bounds, all three axes
		const int NB = 16;
		float bestCost = std::numeric_limits<float>::max(); int bestAxis = -1, bestBin = -1;
		for (int a = 0; a < 3; a++)
		{
			float lo = cb.mn[a], hi = cb.mx[a];
			if (!(hi > lo)) continue;
			B bins[NB]; int cnt[NB];
			for (int k = 0; k < NB; k++) { bins[k] = emptyB(); cnt[k] = 0; }
			float scale = NB / (hi - lo);
			for (int i = start; i < end; i++)
			{
				int k = (int)((cen[3 * idx[i] + a] - lo) * scale); if (k >= NB) k = NB - 1; if (k < 0) k = 0;
				grow(bins[k], pb[idx[i]]); cnt[k]++;
			}
			float rightArea[NB]; int rightCnt[NB];
			B acc = emptyB(); int c = 0;
			for (int k = NB - 1; k > 0; k--) { grow(acc, bins[k]); c += cnt[k]; rightArea[k] = area(acc); rightCnt[k] = c; }
			acc = emptyB(); c = 0;
			for (int k = 0; k < NB - 1; k++)
			{
				grow(acc, bins[k]); c += cnt[k];
				if (c == 0 || rightCnt[k + 1] == 0) continue;
				float cost = area(acc) * c + rightArea[k + 1] * rightCnt[k + 1];
				if (cost < bestCost) { bestCost = cost; bestAxis = a; bestBin = k; }
			}
		}
		float leafCost = area(nb) * n;
		int mid = -1;
		if (bestAxis >= 0 && (n > maxLeaf || bestCost + area(nb) * travCost < leafCost))
		{
			float lo = cb.mn[bestAxis], hi = cb.mx[bestAxis]; float scale = NB / (hi - lo);
			int a = bestAxis, bb = bestBin;
			int32_t* p = std::partition(idx.data() + start, idx.data() + end, [&](int32_t i) {
				int k = (int)((cen[3 * i + a] - lo) * scale); if (k >= NB) k = NB - 1; if (k < 0) k = 0; return k <= bb; });
			mid = (int)(p - idx.data());
		}
		if (mid <= start || mid >= end)
		{
			if (n <= maxLeaf) { leaf(node, start, end); return node; }
			// degenerate centroids: median split on the widest node axis
			int a = 0; float w = -1; for (int k = 0; k < 3; k++) if (nb.mx[k] - nb.mn[k] > w) { w = nb.mx[k] - nb.mn[k]; a = k; }
			mid = start + n / 2;
			std::nth_element(idx.begin() + start, idx.begin() + mid, idx.begin() + end, [&](int32_t x, int32_t y) { return cen[3 * x + a] < cen[3 * y + a]; });
		}
		int l, r;
		if (fork > 0 && n >= 8192)
		{
			FlatBVH lo, ro;
			Builder lb(pb, cen, idx, lo, maxLeaf), rb(pb, cen, idx, ro, maxLeaf);
			lb.travCost = rb.travCost = travCost;
			std::future<int> fl = std::async(std::launch::async, [&lb, start, mid, fork]() { return lb.build(start, mid, fork - 1); });
			rb.build(mid, end, fork - 1);
			fl.get();
			l = append(lo); r = append(ro);
		}
		else { l = build(start, mid, fork); r = build(mid, end, fork); }
		out.left[node] = l; out.right[node] = r;
		return node;
	}
};
}

void BuildBVH(const std::vector<FBounds3>& primBounds, FlatBVH& out, int maxLeaf)
{
	out = FlatBVH();
	if (primBounds.empty()) return;
	std::vector<B> pb(primBounds.size());
	for (size_t i = 0; i < pb.size(); i++)
	{
		pb[i].mn[0] = primBounds[i]._min.x; pb[i].mn[1] = primBounds[i]._min.y; pb[i].mn[2] = primBounds[i]._min.z;
		pb[i].mx[0] = primBounds[i]._max.x; pb[i].mx[1] = primBounds[i]._max.y; pb[i].mx[2] = primBounds[i]._max.z;
	}
	std::vector<float> cen(pb.size() * 3); std::vector<int32_t> idx(pb.size());
	for (size_t i = 0; i < pb.size(); i++) { idx[i] = (int32_t)i; for (int a = 0; a < 3; a++) cen[3 * i + a] = 0.5f * (pb[i].mn[a] + pb[i].mx[a]); }
	int fork = 4;                                                  // up to 16 concurrent subtree builds
	if (const char* e = getenv("JETPBRT_BVH_THREADS")) { int v = atoi(e); fork = v <= 1 ? 0 : (v <= 2 ? 1 : (v <= 4 ? 2 : (v <= 8 ? 3 : 4))); }
	Builder b(pb, cen, idx, out, maxLeaf);
	if (const char* e = getenv("JETPBRT_SAH_TRAV_COST")) { const float v = (float)atof(e); if (v > 0.f && v <= 64.f) b.travCost = v; }
	b.build(0, (int)pb.size(), fork);
}

// ---- the reference's own tree, node for node -------------------------------------------------------------------------
// On finely tessellated meshes the reference's closest hit depends on its tree (DESIGN.md "Numerics": fp32 acceptance
// fringe outside triangle boxes, dropped subtrees), so reproducing its film bit for bit there takes ITS tree: FBVH_Node
// (bvh.h:54-91) draws the split axis with random_int(0, 2) = (int)(0 + 3 * (rand() * (1.0f / (RAND_MAX + 1.0f))))
// (pbrt.h:106-120), std::sorts the range by bounds.min[axis], splits at the median and stops at <= 5 objects; a
// reference process never calls srand, so the sequence is glibc's rand() from its default seed 1 -- restated here
// (TYPE_3 additive feedback generator, r[i] = r[i-3] + r[i-31], 310 outputs discarded) so that the build neither depends on
// nor disturbs the process-wide rand() state.  std::sort is the same libstdc++ introsort the reference build uses, and it
// is applied to the same sequence of (sub)ranges, so equal keys land where they land in the reference.
// An interior node whose only child is a leaf (span <= 5) is emitted as that leaf: its box is the leaf's box.
namespace
{
struct GlibcRand
{
	int32_t r[34]; int f, b;                                                  // f = front, b = rear of the 31-word state
	explicit GlibcRand(uint32_t seed)
	{
		int32_t st[31];
		st[0] = (int32_t)(seed ? seed : 1u);
		for (int i = 1; i < 31; i++)
		{
			long hi = st[i - 1] / 127773, lo = st[i - 1] % 127773;              // 16807 * x mod (2^31 - 1) without overflow
			long w = 16807 * lo - 2836 * hi;
			if (w < 0) w += 2147483647;
			st[i] = (int32_t)w;
		}
		for (int i = 0; i < 31; i++) r[i] = st[i];
		f = 3; b = 0;
		for (int i = 0; i < 310; i++) next();
	}
	int next()
	{
		uint32_t v = (uint32_t)r[f] + (uint32_t)r[b];
		r[f] = (int32_t)v;
		const int result = (int)(v >> 1);
		if (++f >= 31) f = 0;
		if (++b >= 31) b = 0;
		return result;
	}
};

struct RefBuilder
{
	const std::vector<B>& pb; std::vector<int32_t> order; FlatBVH& out; GlibcRand rng;
	RefBuilder(const std::vector<B>& pb, FlatBVH& out) : pb(pb), out(out), rng(1) { order.resize(pb.size()); for (size_t i = 0; i < pb.size(); i++) order[i] = (int32_t)i; }
	int random_axis()
	{
		const float r = (float)rng.next() * (1.0f / (2147483647 + 1.0f));    // pbrt.h:106-108, RAND_MAX = 2147483647
		const float v = 0.0f + (3.0f - 0.0f) * r;                                // pbrt.h:110-114
		return (int)v;                                                           // pbrt.h:116-120
	}
	int emit(const B& b)
	{
		int n = (int)out.left.size();
		out.left.push_back(0); out.right.push_back(0);
		for (int a = 0; a < 3; a++) out.bounds.push_back(b.mn[a]);
		for (int a = 0; a < 3; a++) out.bounds.push_back(b.mx[a]);
		return n;
	}
	void setBox(int n, const B& b) { for (int a = 0; a < 3; a++) { out.bounds[6 * n + a] = b.mn[a]; out.bounds[6 * n + 3 + a] = b.mx[a]; } }
	int build(size_t start, size_t end, B& boxOut)
	{
		const int axis = random_axis();                                          // drawn for every node, leaves included (bvh.h:61)
		const size_t span = end - start;
		const int me = emit(emptyB());
		if (span <= 5)                                                           // MAX_HITTABLES_IN_LEAF
		{
			B bb = emptyB();
			for (size_t i = start; i < end; i++) grow(bb, pb[order[i]]);
			out.left[me] = -((int32_t)out.prim_index.size()) - 1; out.right[me] = (int32_t)span;
			for (size_t i = start; i < end; i++) out.prim_index.push_back(order[i]);
			setBox(me, bb); boxOut = bb;
			return me;
		}
		const std::vector<B>& boxes = pb;
		std::sort(order.begin() + start, order.begin() + end, [&boxes, axis](int32_t x, int32_t y) { return boxes[x].mn[axis] < boxes[y].mn[axis]; });
		const size_t mid = start + span / 2;
		B bl, br;
		const int l = build(start, mid, bl);
		const int r = build(mid, end, br);
		out.left[me] = l; out.right[me] = r;
		B bb = bl; grow(bb, br);
		setBox(me, bb); boxOut = bb;
		return me;
	}
};
}

void BuildReferenceBVH(const std::vector<FBounds3>& primWorldBounds, FlatBVH& out)
{
	out = FlatBVH();
	if (primWorldBounds.empty()) return;
	std::vector<B> pb(primWorldBounds.size());
	for (size_t i = 0; i < pb.size(); i++)
	{
		pb[i].mn[0] = primWorldBounds[i]._min.x; pb[i].mn[1] = primWorldBounds[i]._min.y; pb[i].mn[2] = primWorldBounds[i]._min.z;
		pb[i].mx[0] = primWorldBounds[i]._max.x; pb[i].mx[1] = primWorldBounds[i]._max.y; pb[i].mx[2] = primWorldBounds[i]._max.z;
	}
	RefBuilder b(pb, out);
	B root; b.build(0, pb.size(), root);
}

} // namespace jetpbrt
