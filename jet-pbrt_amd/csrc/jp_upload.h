// jet-pbrt_amd/csrc/jp_upload.h -- host runtime, part 2 of 3: jp_upload_scene -- validation of every index on the host, the device tables (primitive records in
// leaf order, binary / 8-wide / 4-wide trees, the certified walk's tree over the caller's leaves), the device-side hierarchy build (jp_lbvh.h, jp_ploc.h).
// Included by jp_kernels.hip after jp_runtime.h.
#pragma once
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
	const JpOptions& op = c->opt;
	const float extra_pad = std::max(0.f, op.box_pad);
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
		if (op.traversal > 0) { const int m = op.traversal - 1; if (m == 3 && s->bvh_left[0] >= 0) use_wide = true; else if (m >= 0 && m <= 2) use_wide = false; }
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
			uint8_t ql[3][4], qh[3][4]; uint32_t refs[4] = { 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu }, valid = 0, flags = 0;   // unused slots: the empty box (255 > 0) and primitive 0 as a one-primitive leaf (-1): WalkerQ4 tests no valid bit
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
	use_q4 = use_q4 && opt_flag(op.q4, true);
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
	const bool want_cert = s->bvh_reference_semantics == 2 && op.certified >= 0;   // (JpOptions::certified = -1 downgrades to the verbatim walk; nothing upgrades a scene that asked for it)
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
		const float tau = op.cert_eye_tau == 0.f ? 5e-3f : std::max(0.f, op.cert_eye_tau);
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
			const float K = op.cert_slack == 0.f ? 16384.f : std::max(0.f, op.cert_slack);
			cert_pad = (float)(K * 1.1920929e-7 / std::max(1e-20, diag / ni));
			// rays from the camera position: their noise planes are covered by the edge-on flags, so the slack only has to cover the fringe in front of a leaf's box
			const float Ke = op.cert_slack_eye == 0.f ? std::min(K, 1024.f) : std::max(0.f, op.cert_slack_eye);
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
		bool ploc = op.device_tree != 2;
		int maxLeaf = ploc ? 2 : 3;
		if (op.bvh_max_leaf >= 1 && op.bvh_max_leaf <= 16) maxLeaf = op.bvh_max_leaf;
		// [round 3] PLOC clustering (jp_ploc.h) instead of the Karras topology; JETPBRT_DEVICE_TREE=lbvh restores the latter, which also serves
		// as the fallback should the clustering not finish within its round limit
		if (e == hipSuccess && ploc) { e = ploc_build(c->stream, (const float4*)d_p0, (const int4*)d_m0, s->n_primitives, maxLeaf, op.ploc_radius, op.ploc_max_rounds, lr, sorted); if (e == hipErrorNotReady) { e = hipSuccess; ploc = false; } }
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
		const bool want_wide = opt_flag(op.device_wide, s->n_primitives > 64);
		if (want_wide)
		{
			WideResult wr;
			e = lbvh_build_wide(c->stream, (const float4*)c->d_nodes, s->n_primitives, wr);
			if (e != hipSuccess) return fail(JP_ERR_DEVICE, std::string("jp_upload_scene: device wide-tree build failed: ") + hipGetErrorString(e));
			if (wr.d_wide) { c->d_wide = wr.d_wide; dev_wide = true; dev_n_wide = wr.n_wide; wide_height = wr.height; use_wide = true; c->build_ms += wr.build_ms; }
		}
		// [round 3] ... and the 4-wide tree of Walker<4> for the closest-hit (and shadow) rays, as the host path has it
		const bool want_q4 = s->n_primitives > 1024 && opt_flag(op.q4, true);
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

	SceneView& v = c->sv;
	v.nodes = (const float4*)c->d_nodes; v.n_nodes = (int)(n4nodes / 4);
	v.prims = (const float4*)c->d_prims; v.meta = (const int4*)c->d_meta; v.n_prims = (int)nmeta;
	v.mats = (const float4*)c->d_mats; v.mat_type = (const int*)c->d_mat_type; v.n_mats = s->n_materials;
	v.lights = (const float4*)c->d_lights; v.n_lights = s->n_lights; v.shade_tab = (const float4*)c->d_shade_tab;
	v.env_sum = make_float3(envsum[0], envsum[1], envsum[2]); v.n_env = nenv;
	v.world_radius = s->world_radius; v.cam = s->camera;
	v.flat = (const float4*)c->d_flat; v.n_flat = (int)(flat.size() / 2);
	v.wide = (const uint4*)c->d_wide; v.n_wide = dev_wide ? dev_n_wide : (int)(wide.size() / 20);
	v.q4 = (const uint4*)c->d_q4; v.n_q4 = dev_q4 ? dev_n_q4 : (int)(q4.size() / 16);
	v.refbox = (const float4*)c->d_refbox; v.cert_pad = cert_pad; v.cert_pad_eye = cert_pad_eye; c->cert = use_cert;
	c->use_q4 = use_q4; c->q4_shadow = use_q4;                       // shadow rays too (measured against the 8-wide tree: k_shadow 53.8 -> 52.7 ms per 512 spp, frame +4 %)
	c->q4_shadow = use_q4 && opt_flag(op.q4_shadow, true);
	c->stack_depth = std::max(2, height + 2);                         // binary / 8-wide / verbatim walks: the tree's height
	c->stack_depth_q4 = (use_q4 || use_cert) ? 3 * q4_height + 2 : 0;   // 4-wide walks (Walker<4> / <6>): a node pushes up to three children
	size_t scene_bytes = (n4nodes + n4prims) / 4 * 5 * sizeof(float4);   // 80-byte LDS record stride
	size_t prim_bytes = n4prims / 4 * 5 * sizeof(float4);
	size_t stack_bytes = (size_t)c->stack_depth * JP_BLOCK * sizeof(int);
	c->scene_in_lds = !device_build && scene_bytes + stack_bytes <= 40 * 1024;   // device-built trees are indexed sparsely (Karras numbering): global memory only
	c->trav_mode = use_wide ? 3 : ((!flat.empty() && prim_bytes <= 40 * 1024) ? 2 : (c->scene_in_lds ? 1 : 0));
	if (!use_wide && op.traversal > 0) { const int m = op.traversal - 1; if (m == 0 || (m == 1 && c->scene_in_lds)) c->trav_mode = m; }   // experiments: force a lower mode
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
		c->stack_lds_words = op.stack_lds_words >= 2 ? (op.stack_lds_words & ~1) : 12;   // even: the wide tree's entries are word pairs
		// lane refill in the traversal kernels (k_extend_persist / k_shadow_persist): on by default for scenes walked through global
		// memory (measured on the 280k-triangle scene: k_extend 39.1 -> 28.4 ms, k_shadow 28.8 -> 18.9 ms per 128 spp; reference-tree
		// mode 154 -> 227 Msamples/s); the LDS-resident Cornell box loses with it (reference-tree mode 1109 -> 965), so small scenes keep
		// the one-ray-per-lane kernels.  JETPBRT_PERSIST = 0 (off) or the refill threshold (8 / 16 / 32 idle lanes).
		c->persist = ((c->trav_mode == 0 || c->trav_mode == 3 || c->trav_mode == 5) && s->n_primitives > 1024) ? 16 : 0;
		if (op.persist != 0) c->persist = op.persist < 0 ? 0 : op.persist;
		// each iteration the lanes of a wave vote on the kind of step it runs (node / leaf); measured on the 280k-triangle scene: k_extend
		// 28.3 -> 21.9 ms, k_shadow 18.9 -> 16.5 ms per 128 spp.  The reference-tree walk (one node per step, leaf objects as their own
		// steps) is faster without it: 310 vs 286 Msamples/s.
		c->vote = opt_flag(op.vote, c->trav_mode != 5 || c->cert);
		c->shade_sort = nk > 1;
		c->class_mask = 1; for (int k = 0; k < 5; k++) if (kinds[k]) c->class_mask |= 2 << k;
		c->shade_sort = opt_flag(op.shade_sort, c->shade_sort);
	}
	c->has_null_material = hasNull; c->cert_fell_back = false;
	c->have_scene = true;
	return JP_OK;
}

