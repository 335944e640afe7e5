// jet-pbrt_amd/csrc/jp_device.h -- device-side math, scene tables, BVH traversal and shape intersection.
//
// Parity rules (DESIGN.md "Numerics"): everything that restates reference arithmetic is plain fp32 in the
// reference's operation order; this file is compiled with -ffp-contract=off so no mul+add pair is fused, and
// fp32 divide / sqrt are the correctly rounded forms (hipcc default).  Only the BVH slab test -- our own
// topology, conservative by construction -- uses explicit fmaf / min3 / max3.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "jetpbrt_amd.h"
#include "jp_counter_rng.h"

#define JP_BLOCK 256
#define JP_STACK_DEPTH 32

namespace jp
{
struct V3 { float x, y, z; };
__device__ __forceinline__ V3 mk(float x, float y, float z) { V3 v; v.x = x; v.y = y; v.z = z; return v; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 operator/(V3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ V3 cmul(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 cdiv(V3 a, V3 b) { return mk(a.x / b.x, a.y / b.y, a.z / b.z); }
__device__ __forceinline__ V3 splat(float v) { return mk(v, v, v); }
__device__ __forceinline__ V3 csqrt(V3 a) { return mk(sqrtf(a.x), sqrtf(a.y), sqrtf(a.z)); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }          // geometry.h:107
__device__ __forceinline__ float len2(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }               // geometry.h:101
__device__ __forceinline__ float len(V3 a) { return sqrtf(len2(a)); }                              // geometry.h:102
__device__ __forceinline__ V3 normalize(V3 a) { return a / len(a); }                                    // geometry.h:104
__device__ __forceinline__ V3 cross(V3 a, V3 v) { return mk(a.y * v.z - a.z * v.y, a.z * v.x - a.x * v.z, a.x * v.y - a.y * v.x); }
__device__ __forceinline__ float absdot(V3 a, V3 b) { return fabsf(dot(a, b)); }
__device__ __forceinline__ bool isblack(V3 c) { return c.x == 0.f && c.y == 0.f && c.z == 0.f; }        // color.h:50
// std::min / std::max with libstdc++'s argument semantics (NaN-order sensitive)
__device__ __forceinline__ float smin(float a, float b) { return (b < a) ? b : a; }
__device__ __forceinline__ float smax(float a, float b) { return (a < b) ? b : a; }
__device__ __forceinline__ float maxcomp(V3 c) { float m = (c.y < c.z) ? c.z : c.y; return (c.x < m) ? m : c.x; }   // color.h:40-43
__device__ __forceinline__ float clampf(float v, float lo, float hi) { if (v < lo) return lo; else if (v > hi) return hi; else return v; }   // pbrt.h:73-83
__device__ __forceinline__ V3 xyz(float4 v) { return mk(v.x, v.y, v.z); }

#define JP_PI      3.14159274101257324219f          /* (float)3.14159265358979323846, pbrt.h:39 */
#define JP_INF     __int_as_float(0x7f800000)

// ---- device scene tables ---------------------------------------------------------------------------------------
// Primitive record, 4 x float4 (64 B), stored in BVH leaf order:
//   triangle : g0 = (p0, -), g1 = (p1, -), g2 = (p2, -),   g3 = (n, type)
//   rectangle: g0 = (p0, p3.x), g1 = (p1, p3.y), g2 = (p2, p3.z), g3 = (n, type)
//   sphere   : g0 = (c, r),                                 g3 = (-, -, -, type)
// meta = (orig primitive id, material, light, shape type)
// BVH node, 4 x float4 (64 B): both children's boxes live in the parent so one fetch decides both descents:
//   n0 = (L.min.xyz, L.max.x)  n1 = (L.max.yz, R.min.xy)  n2 = (R.min.z, R.max.xyz)  n3 = (ref L, ref R, -, -) as int bits
//   ref >= 0: interior node index; ref < 0: leaf, e = -ref-1, first = e >> 4, count = (e & 15) + 1
struct SceneView
{
	const float4* nodes; int n_nodes;
	const float4* prims; const int4* meta; int n_prims;
	const float4* mats;                     // 4 x float4 per material (JP_MAT_PARAM_STRIDE floats)
	const int* mat_type; int n_mats;
	const float4* lights;                   // 2 x float4 per light: (radiance, type bits), then AREA (device prim bits, 1/area, -, -), POINT / DIRECTION (vec xyz, -)
	int n_lights;
	const float4* shade_tab;                // k_shade's LDS tables as one array in LDS order: lights | mats | mat_type (padded to 16 B) [| prims | meta]
	float3 env_sum;                         // sum of the infinite lights' radiance in Lights() order (light.h:300-303)
	int n_env;
	float world_radius;
	JpCamera cam;
	const float4* flat; int n_flat;        // tiny scenes: <= 32 leaf boxes with the bit set of their primitives (<= 64), flat_boxes
	const uint4* wide; int n_wide;          // large scenes: 8-wide quantised nodes, 5 x 16 bytes each (traverse_wide)
	const uint4* q4; int n_q4;              // large scenes: 4-wide tree with quantised child boxes, 4 x 16 bytes per node (Walker<4>)
	const float4* refbox; float cert_pad, cert_pad_eye;   // (cert_pad_eye: the slack for rays from the camera position, which the edge-on flags cover)  // reference semantics, certified walk (Walker<6>): per device primitive the exact box of its leaf in the caller's tree (min, max); distance-cull slack c in tmax + c * tmax^2
};

// ---- shape intersection: exact restatements ---------------------------------------------------------------------
// FTriangle::Intersect shape.h:291-327.  On acceptance `tmax` shrinks (ray.SetMaxT).
__device__ __forceinline__ bool tri_hit(V3 p0, V3 p1, V3 p2, V3 n, V3 o, V3 d, float tmin, float& tmax)
{
	// the reference evaluates the three edge functions first and the plane distance second; both are pure, so the
	// cheap test goes first here: most candidates fail the (tmin, tmax) interval and never pay for the cross products
	const V3 oa = p0 - o;
	const float distance = dot(n, oa) / dot(n, d);
	if (!((distance > tmin) && (distance < tmax))) return false;
	const V3 ob = p1 - o, oc = p2 - o;
	const V3 v0 = cross(oc, ob), v1 = cross(ob, oa), v2 = cross(oa, oc);
	const float v0d = dot(v0, d), v1d = dot(v1, d), v2d = dot(v2, d);
	if (((v0d < 0) && (v1d < 0) && (v2d < 0)) || ((v0d >= 0) && (v1d >= 0) && (v2d >= 0))) { tmax = distance; return true; }
	return false;
}
// FRectangle::Intersect shape.h:399-435
__device__ __forceinline__ bool rect_hit(V3 p0, V3 p1, V3 p2, V3 p3, V3 n, V3 o, V3 d, float tmin, float& tmax)
{
	const V3 oa = p0 - o;
	const float distance = dot(n, oa) / dot(n, d);
	if (!((distance > tmin) && (distance < tmax))) return false;
	const V3 ob = p1 - o, oc = p2 - o, od = p3 - o;
	const V3 v0 = cross(oc, ob), v1 = cross(ob, oa), v2 = cross(oa, od), v3 = cross(od, oc);
	const float v0d = dot(v0, d), v1d = dot(v1, d), v2d = dot(v2, d), v3d = dot(v3, d);
	if (((v0d < 0) && (v1d < 0) && (v2d < 0) && (v3d < 0)) || ((v0d >= 0) && (v1d >= 0) && (v2d >= 0) && (v3d >= 0))) { tmax = distance; return true; }
	return false;
}
// FSphere::Intersect shape.h:487-526 (sqrt resolves to the double overload there; for sqrt the double rounding
// is innocuous, so the correctly rounded fp32 sqrt gives the identical value)
__device__ __forceinline__ bool sph_hit(V3 c, float r, V3 o, V3 d, float tmin, float& tmax)
{
	V3 oc = o - c;
	float a = len2(d);
	float half_b = dot(oc, d);
	float cc = len2(oc) - r * r;
	float disc = half_b * half_b - a * cc;
	if (disc > 0.0f)
	{
		float root = sqrtf(disc);
		float time;
		float root1 = (-half_b - root) / a;
		if (root1 < tmax && root1 > tmin) time = root1;
		else
		{
			float root2 = (-half_b + root) / a;
			if (root2 < tmax && root2 > tmin) time = root2;
			else return false;
		}
		tmax = time;
		return true;
	}
	return false;
}

// FDisk::Intersect shape.h:200-221 (isEqual of pbrt.h:97-104 with epsilon = FLT_EPSILON, Distance = sqrt of the squared length)
__device__ __forceinline__ bool disk_hit(V3 c, float r, V3 n, V3 o, V3 d, float tmin, float& tmax)
{
	const float dn = dot(d, n);
	if (fabsf(dn - 0.f) <= 1.1920928955078125e-07f * smax(1.f, smax(fabsf(dn), fabsf(0.f)))) return false;
	const V3 op = c - o;
	const float distance = dot(n, op) / dot(n, d);
	if ((distance > tmin) && (distance < tmax))
	{
		const V3 hp = o + distance * d;
		if (len(c - hp) <= r) { tmax = distance; return true; }
	}
	return false;
}

// One primitive record against the ray; FPrimitive::Intersect primitive.h:39-48.  kS = record stride in float4
// units: 4 in global memory, 5 in LDS (the 80-byte stride spreads randomly indexed 64-byte records over all
// bank groups instead of four).
template <int kS, typename PrimPtr>
__device__ __forceinline__ bool prim_hit(PrimPtr prims, int pi, V3 o, V3 d, float tmin, float& tmax)
{
	// all four quads of the record are requested together (no load waits on the shape type): one memory latency per
	// primitive instead of two
	const float4 g0 = prims[kS * pi + 0], g1 = prims[kS * pi + 1], g2 = prims[kS * pi + 2], g3 = prims[kS * pi + 3];
	const int type = __float_as_int(g3.w);
	if (type == JP_SHAPE_TRIANGLE) return tri_hit(xyz(g0), xyz(g1), xyz(g2), xyz(g3), o, d, tmin, tmax);
	if (type == JP_SHAPE_SPHERE) return sph_hit(xyz(g0), g0.w, o, d, tmin, tmax);
	if (type == JP_SHAPE_DISK) return disk_hit(xyz(g0), g0.w, xyz(g1), o, d, tmin, tmax);
	return rect_hit(xyz(g0), xyz(g1), xyz(g2), mk(g0.w, g1.w, g2.w), xyz(g3), o, d, tmin, tmax);
}

// ---- BVH traversal (replaces FBVH_Node::Intersect bvh.h:94-103: ordered, early-out, any-hit for shadows) -------
// Driver: one ray per lane.  `stack` is this thread's column of the LDS stack (entry k at stack[k * JP_BLOCK]).
// "while-while" form: a lane walks interior nodes until it holds a leaf, then intersects the leaf.  Both children's
// slabs are tested from one 64-byte node with t = (b - o) * (1/d): relative error of a few ulp in t (the
// fma(b, 1/d, -o/d) form would lose absolute accuracy for near-axis-parallel rays); boxes are padded at upload and
// compared with 2e-6 slack, so our own topology may be tested more generously than the reference's boxes, never more
// strictly than the geometry.  fminf/fmaxf drop the NaN of 0 * inf (ray lying in a slab plane).
// Returns the device primitive index of the accepted hit (-1: none); `tmax` = hit distance.
template <bool kAnyHit, int kS, typename NodePtr, typename PrimPtr>
__device__ __forceinline__ int traverse(NodePtr nodes, PrimPtr prims, V3 o, V3 d, float tmin, float& tmax, int* stack)
{
	const float ix = 1.0f / d.x, iy = 1.0f / d.y, iz = 1.0f / d.z;
	int hit = -1, sp = 0, cur = 0;                                // node 0 is the root (always interior on the device)
	for (;;)
	{
		bool alive = true;
		while (cur >= 0)
		{
			const float4 n0 = nodes[kS * cur + 0], n1 = nodes[kS * cur + 1], n2 = nodes[kS * cur + 2], n3 = nodes[kS * cur + 3];
			const float lx0 = (n0.x - o.x) * ix, lx1 = (n0.w - o.x) * ix;
			const float ly0 = (n0.y - o.y) * iy, ly1 = (n1.x - o.y) * iy;
			const float lz0 = (n0.z - o.z) * iz, lz1 = (n1.y - o.z) * iz;
			const float ln = fmaxf(fmaxf(fminf(lx0, lx1), fminf(ly0, ly1)), fmaxf(fminf(lz0, lz1), tmin));
			const float lf = fminf(fminf(fmaxf(lx0, lx1), fmaxf(ly0, ly1)), fminf(fmaxf(lz0, lz1), tmax));
			const float rx0 = (n1.z - o.x) * ix, rx1 = (n2.y - o.x) * ix;
			const float ry0 = (n1.w - o.y) * iy, ry1 = (n2.z - o.y) * iy;
			const float rz0 = (n2.x - o.z) * iz, rz1 = (n2.w - o.z) * iz;
			const float rn = fmaxf(fmaxf(fminf(rx0, rx1), fminf(ry0, ry1)), fmaxf(fminf(rz0, rz1), tmin));
			const float rf = fminf(fminf(fmaxf(rx0, rx1), fmaxf(ry0, ry1)), fminf(fmaxf(rz0, rz1), tmax));
			const bool hl = ln <= lf * 1.000002f, hr = rn <= rf * 1.000002f;
			const int cl = __float_as_int(n3.x), cr = __float_as_int(n3.y);
			if (hl && hr)
			{
				const bool leftFirst = kAnyHit ? true : (ln <= rn);
				cur = leftFirst ? cl : cr;
				stack[sp * JP_BLOCK] = leftFirst ? cr : cl; sp++;
			}
			else if (hl) cur = cl;
			else if (hr) cur = cr;
			else if (sp > 0) { sp--; cur = stack[sp * JP_BLOCK]; }
			else { alive = false; break; }
		}
		if (!alive) break;
		const int e = -cur - 1, first = e >> 4, count = (e & 15) + 1;
		for (int k = 0; k < count; k++)
			if (prim_hit<kS>(prims, first + k, o, d, tmin, tmax)) { hit = first + k; if (kAnyHit) return hit; }
		if (sp > 0) { sp--; cur = stack[sp * JP_BLOCK]; } else break;
	}
	return hit;
}

// REFERENCE SEMANTICS (JpScene.bvh_reference_semantics): the caller's tree walked node for node the way the reference does.
// FBVH_Node::Intersect bvh.h:94-103: box test on the node's own unpadded bounds, then the left subtree, then the right one,
// no ordering by distance; FBVH_NodeLeaf::Intersect bvh.h:132-142: every object of the leaf in order; every accepted hit
// shrinks ray.max_t, which later box tests see.  FBounds3::Intersect geometry.cc:10-30 verbatim: per axis two IEEE divisions,
// std::min / std::max argument order, `tmax <= tmin` rejects.  With the reference's own tree this returns the reference's hit
// even where that depends on the topology (hits in the fp32 acceptance fringe outside a triangle's box, subtrees dropped by
// the strict box test).
// FBounds3::Intersect geometry.cc:10-30, verbatim: per axis two IEEE divisions, std::min / std::max argument order (NaN-order
// sensitive), `tmax <= tmin` rejects after every axis.
__device__ __forceinline__ bool ref_box_exact(float4 n0, float4 n1, V3 o, V3 d, float tmin, float tmax)
{
	float bt0 = tmin, bt1 = tmax;
	{
		const float lo = (n0.x - o.x) / d.x, hi = (n1.x - o.x) / d.x;
		bt0 = smax(smin(lo, hi), bt0); bt1 = smin(smax(lo, hi), bt1);
		if (bt1 <= bt0) return false;
	}
	{
		const float lo = (n0.y - o.y) / d.y, hi = (n1.y - o.y) / d.y;
		bt0 = smax(smin(lo, hi), bt0); bt1 = smin(smax(lo, hi), bt1);
		if (bt1 <= bt0) return false;
	}
	{
		const float lo = (n0.z - o.z) / d.z, hi = (n1.z - o.z) / d.z;
		bt0 = smax(smin(lo, hi), bt0); bt1 = smin(smax(lo, hi), bt1);
		if (bt1 <= bt0) return false;
	}
	return true;
}
// The same decision, mostly without the six divisions.  Without NaNs the reference's verdict is  min(far_x, far_y, far_z, tmax)
// > max(near_x, near_y, near_z, tmin)  (the running interval only shrinks, so the per-axis early rejects are implied by the final
// one).  With t~ = (b - o) * rcp(d) every slab distance carries a relative error below 4e-7 (fp32 subtraction identical, 1-ulp
// reciprocal, one rounding), so when the two sides are further apart than 2e-6 of their magnitudes the verdict is the exact
// verdict; otherwise -- or when any operand is not finite (axis-parallel rays: 0 * inf, inf - inf) -- the exact test decides.
__device__ __forceinline__ bool ref_box(float4 n0, float4 n1, V3 o, V3 d, V3 rd, float tmin, float tmax)
{
	const float x0 = (n0.x - o.x) * rd.x, x1 = (n1.x - o.x) * rd.x;
	const float y0 = (n0.y - o.y) * rd.y, y1 = (n1.y - o.y) * rd.y;
	const float z0 = (n0.z - o.z) * rd.z, z1 = (n1.z - o.z) * rd.z;
	const float t0 = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), tmin));
	const float t1 = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), tmax));
	const float s = x0 + x1 + y0 + y1 + z0 + z1;                  // NaN or inf anywhere -> not finite (inf - inf -> NaN)
	const float margin = 2e-6f * (fabsf(t0) + fabsf(t1));
	const float gap = t1 - t0;
	if (fabsf(s) <= 3.0e38f && fabsf(gap) > margin) return gap > 0.f;
	return ref_box_exact(n0, n1, o, d, tmin, tmax);
}

// nodes: 2 x float4 per node, (min xyz, left bits) (max xyz, right bits); left < 0: leaf with device primitives [-(left) - 1, + right).
// (Tried and dropped: both children's boxes in the parent's record, one fetch per interior node and no fetch for rejected children,
// with the right child's early verdict kept only while no hit was accepted in between -- bit-identical, but the unordered walk
// accepts hits all the time, so right boxes get tested twice: k_extend 51 -> 66 ms on the 280k-triangle scene.)
template <bool kAnyHit>
__device__ __forceinline__ int traverse_ref(const float4* __restrict__ nodes, const float4* __restrict__ prims, V3 o, V3 d, float tmin, float& tmax, int* stack)
{
	const V3 rd = mk(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));
	int hit = -1, sp = 0, cur = 0;
	for (;;)
	{
		const float4 n0 = nodes[2 * cur], n1 = nodes[2 * cur + 1];
		const bool ok = ref_box(n0, n1, o, d, rd, tmin, tmax);
		const int left = __float_as_int(n0.w), right = __float_as_int(n1.w);
		if (ok && left >= 0) { stack[sp * JP_BLOCK] = right; sp++; cur = left; continue; }
		if (ok)
		{
			const int first = -left - 1;
			for (int k = 0; k < right; k++)
				if (prim_hit<4>(prims, first + k, o, d, tmin, tmax)) { hit = first + k; if (kAnyHit) return hit; }
		}
		if (sp == 0) break;
		sp--; cur = stack[sp * JP_BLOCK];
	}
	return hit;
}

// Tiny scenes (<= 32 BVH leaves holding <= 64 primitives, e.g. the 32-triangle Cornell box): the tree is collapsed into ONE
// wide node whose children are the leaves.  Phase 1 (flat_boxes) tests every leaf box with wave-UNIFORM control flow and
// uniform (scalar / broadcast) operands -- no stack, no pointer chasing, all 64 lanes busy -- and records the PRIMITIVES of
// the boxes the ray enters in a per-lane bit mask (one bit per device primitive).  Phase 2 (flat_prims) walks the lane's own
// set bits and runs the exact primitive tests.  For the Cornell box this replaces ~6-8 dependent binary-node steps per ray by
// 17 independent slab tests.  The popcount of the mask is the number of primitive tests the ray will pay: the sorted kernels
// (k_extend_flat / k_shadow_flat, jp_kernels.hip) bin the rays of a tile by it so that the lanes of a wave finish together.
// flat[2i] = (box min xyz, primitive bits 0..31), flat[2i+1] = (box max xyz, primitive bits 32..63) -- 8 dwords per leaf, all
// fetched by scalar loads; the loop runs in unrolled groups of four (scalar loads issued together) plus a tail.
// The reciprocal direction is the hardware approximation (1 ulp): our own boxes are padded by 1e-6 relative and compared
// with 2e-6 slack, so the test stays conservative; a zero or denormal component gives +-inf, the axis-parallel case.
typedef const __attribute__((address_space(4))) float* ConstFPtr;       // constant address space: uniform indices become scalar (s_load) loads
typedef unsigned long long u64;
template <bool k64> struct FlatMask { typedef unsigned int type; };
template <> struct FlatMask<true> { typedef u64 type; };
template <bool k64> __device__ __forceinline__ int mask_count(typename FlatMask<k64>::type m) { return k64 ? __popcll((u64)m) : __popc((unsigned int)m); }

template <bool k64>
__device__ __forceinline__ typename FlatMask<k64>::type flat_boxes(const float4* flat_g, int n_flat, V3 o, V3 d, float tmin, float tmax)
{
	typedef typename FlatMask<k64>::type M;
	ConstFPtr flat = (ConstFPtr)flat_g;
	const float ix = __builtin_amdgcn_rcpf(d.x), iy = __builtin_amdgcn_rcpf(d.y), iz = __builtin_amdgcn_rcpf(d.z);
	M mask = 0;
	#define JP_FLAT_BOX(i) \
	{ \
		const float x0 = (flat[8 * (i) + 0] - o.x) * ix, x1 = (flat[8 * (i) + 4] - o.x) * ix; \
		const float y0 = (flat[8 * (i) + 1] - o.y) * iy, y1 = (flat[8 * (i) + 5] - o.y) * iy; \
		const float z0 = (flat[8 * (i) + 2] - o.z) * iz, z1 = (flat[8 * (i) + 6] - o.z) * iz; \
		const float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), tmin)); \
		const float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), tmax)); \
		M bits = (M)__float_as_uint(flat[8 * (i) + 3]); \
		if (k64) bits |= (M)((u64)__float_as_uint(flat[8 * (i) + 7]) << 32); \
		if (tn <= tf * 1.000002f) mask |= bits; \
	}
	int i0 = 0;
	for (; i0 + 4 <= n_flat; i0 += 4)
	{
		#pragma unroll
		for (int u = 0; u < 4; u++) JP_FLAT_BOX(i0 + u)
	}
	for (; i0 < n_flat; i0++) JP_FLAT_BOX(i0)
	#undef JP_FLAT_BOX
	return mask;
}

template <bool kAnyHit, bool k64, int kS, typename PrimPtr>
__device__ __forceinline__ int flat_prims(typename FlatMask<k64>::type mask, PrimPtr prims, V3 o, V3 d, float tmin, float& tmax)
{
	int hit = -1;
#ifdef JP_DBG_FLAT_P1ONLY
	mask = mask == (typename FlatMask<k64>::type)0x12345677u ? 1u : 0u;   // timing experiment: box phase (and sort) only
#endif
	while (mask)
	{
		const int p = (k64 ? __ffsll((unsigned long long)mask) : __ffs((int)mask)) - 1;
		mask &= mask - 1;
		if (prim_hit<kS>(prims, p, o, d, tmin, tmax)) { hit = p; if (kAnyHit) return hit; }
	}
	return hit;
}

template <bool kAnyHit, int kS, typename PrimPtr>
__device__ __forceinline__ int traverse_flat(const float4* flat_g, int n_flat, int n_prims, PrimPtr prims, V3 o, V3 d, float tmin, float& tmax)
{
	if (n_prims > 32) return flat_prims<kAnyHit, true, kS>(flat_boxes<true>(flat_g, n_flat, o, d, tmin, tmax), prims, o, d, tmin, tmax);   // wave-uniform branch
	return flat_prims<kAnyHit, false, kS>(flat_boxes<false>(flat_g, n_flat, o, d, tmin, tmax), prims, o, d, tmin, tmax);
}

// Large scenes: 8-wide BVH with quantised child boxes (after Ylitie, Karras, Laine: "Efficient Incoherent Ray Traversal
// on GPUs Through Compressed Wide BVHs", HPG 2017), 80 bytes per node.  Incoherent rays on a 280k-triangle scene are
// bound by the number of distinct cache lines a wave's loads touch (every lane walks its own node): the binary tree
// costs ~35 node visits x 4 loads per ray, the wide tree ~13 visits x 5 loads.
//   q0 = (p.x, p.y, p.z, e.x | e.y << 8 | e.z << 16 | imask << 24)     node origin, per-axis scale exponents, inner mask
//   q1 = (first child node, first primitive, meta[0..3], meta[4..7])
//   q2 = (lo.x[0..3], lo.x[4..7], lo.y[0..3], lo.y[4..7])   q3 = (lo.z.., lo.z.., hi.x.., hi.x..)   q4 = (hi.y.., hi.y.., hi.z.., hi.z..)
// child box plane = p + q * 2^e (the host rounds q outward until the fp32 value of that expression is conservative).
// meta: 0 empty; inner child: 0x20 | (24 + slot); leaf child: unary primitive count << 5 | offset (offset + count <= 24).
// Children sit in the slot whose three bits say on which side of the node centre they lie, so 24 + (slot ^ (7 - octant))
// orders the inner hits front to back for the ray's direction octant; inner children are stored contiguously in slot order.
// The stack holds (node group base, hit bits | imask) pairs: entry k at stack[2k * JP_BLOCK], stack[(2k+1) * JP_BLOCK].
template <bool kAnyHit, typename PrimPtr>
__device__ __forceinline__ int traverse_wide(const uint4* __restrict__ wide, PrimPtr prims, V3 o, V3 d, float tmin, float& tmax, unsigned int* stack)
{
	const float ix = 1.0f / d.x, iy = 1.0f / d.y, iz = 1.0f / d.z;
	const unsigned int octinv = 7u - ((d.x < 0 ? 1u : 0u) | (d.y < 0 ? 2u : 0u) | (d.z < 0 ? 4u : 0u));
	unsigned int ngx = 0, ngy = 0x80000000u;                      // the root: group base 0, one pending inner hit
	int hit = -1, sp = 0;
	for (;;)
	{
		unsigned int tgx = 0, tgy = 0;
		if (ngy > 0x00ffffffu)
		{
			const unsigned int bit = 31u - (unsigned int)__clz((int)ngy);
			ngy &= ~(1u << bit);
			if (ngy > 0x00ffffffu) { stack[(2 * sp) * JP_BLOCK] = ngx; stack[(2 * sp + 1) * JP_BLOCK] = ngy; sp++; }
			const unsigned int slot = (bit - 24u) ^ octinv;
			const unsigned int rel = (unsigned int)__popc(ngy & 0xffu & ~(0xffffffffu << slot));
			const unsigned int idx = ngx + rel;
			const uint4 q0 = wide[5 * idx + 0], q1 = wide[5 * idx + 1], q2 = wide[5 * idx + 2], q3 = wide[5 * idx + 3], q4 = wide[5 * idx + 4];
			const float sx = __uint_as_float(((q0.w & 0xffu) ) << 23), sy = __uint_as_float(((q0.w >> 8) & 0xffu) << 23), sz = __uint_as_float(((q0.w >> 16) & 0xffu) << 23);
			const float px = __uint_as_float(q0.x) - o.x, py = __uint_as_float(q0.y) - o.y, pz = __uint_as_float(q0.z) - o.z;
			unsigned int hm = 0;
			#pragma unroll
			for (int i = 0; i < 8; i++)
			{
				const int sh = 8 * (i & 3);
				const unsigned int m = ((i < 4 ? q1.z : q1.w) >> sh) & 0xffu;
				const float lx = (float)(((i < 4 ? q2.x : q2.y) >> sh) & 0xffu), ly = (float)(((i < 4 ? q2.z : q2.w) >> sh) & 0xffu), lz = (float)(((i < 4 ? q3.x : q3.y) >> sh) & 0xffu);
				const float hx = (float)(((i < 4 ? q3.z : q3.w) >> sh) & 0xffu), hy = (float)(((i < 4 ? q4.x : q4.y) >> sh) & 0xffu), hz = (float)(((i < 4 ? q4.z : q4.w) >> sh) & 0xffu);
				const float x0 = fmaf(lx, sx, px) * ix, x1 = fmaf(hx, sx, px) * ix;
				const float y0 = fmaf(ly, sy, py) * iy, y1 = fmaf(hy, sy, py) * iy;
				const float z0 = fmaf(lz, sz, pz) * iz, z1 = fmaf(hz, sz, pz) * iz;
				const float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), tmin));
				const float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), tmax));
				if (m != 0u && tn <= tf * 1.000002f)
				{
					const bool inner = (m & 0x18u) == 0x18u;
					const unsigned int bits = inner ? 1u : (m >> 5), at = inner ? 24u + ((m & 7u) ^ octinv) : (m & 31u);
					hm |= bits << at;
				}
			}
			ngx = q1.x; ngy = (hm & 0xff000000u) | (q0.w >> 24);
			tgx = q1.y; tgy = hm & 0x00ffffffu;
		}
		while (tgy)
		{
			const int j = __ffs((int)tgy) - 1;
			tgy &= tgy - 1;
			if (prim_hit<4>(prims, (int)tgx + j, o, d, tmin, tmax)) { hit = (int)tgx + j; if (kAnyHit) return hit; }
		}
		if (ngy <= 0x00ffffffu)
		{
			if (sp == 0) break;
			sp--; ngx = stack[(2 * sp) * JP_BLOCK]; ngy = stack[(2 * sp + 1) * JP_BLOCK];
		}
	}
	return hit;
}

// ---- resumable traversal ("walkers") for the lane-refill kernels (k_extend_persist / k_shadow_persist, jp_kernels.hip) -------------
// The same three traversals as above -- traverse (mode 0), traverse_wide (mode 3), traverse_ref (mode 5) -- cut into steps: one
// step() handles ONE interior node, ONE leaf (mode 0 / 5) or one primitive / one pop (mode 3), so that the lanes of a wave, each at
// its own stage of its own ray, advance together and a finished lane can take the next ray while the others go on.  The order in
// which a ray's nodes and primitives are visited, the arithmetic and the acceptance rules are those of the loops above: identical
// hit records.  `stack` is the thread's column of the LDS stack.  done: the ray is finished (hit >= 0: accepted primitive).
// Traversal stack of the resumable walkers (lane refill kernels): the first `cap` words of a thread's column live in LDS (word w of
// thread t at lds[w * JP_BLOCK]), deeper ones in a global spill array (word w at spill[(w - cap) * stride]).  A stack as deep as the
// tree is high is what the worst case needs and what almost no ray uses; keeping only the shallow part in LDS lets 8 workgroups
// instead of 5-6 share a CU.
struct WalkStack
{
	int* lds; int* spill; int cap; unsigned int stride;
	__device__ __forceinline__ void put(int at, int v) const { if (at < cap) lds[at * JP_BLOCK] = v; else spill[(size_t)(at - cap) * stride] = v; }
	__device__ __forceinline__ int get(int at) const { return at < cap ? lds[at * JP_BLOCK] : spill[(size_t)(at - cap) * stride]; }
};

template <int kMode> struct Walker;

template <> struct Walker<0>
{
	V3 o, d; float ix, iy, iz, tmin, tmax; int cur, sp, hit; bool done;
	__device__ __forceinline__ void start(V3 o_, V3 d_, float tmin_, float tmax_)
	{ o = o_; d = d_; ix = 1.0f / d.x; iy = 1.0f / d.y; iz = 1.0f / d.z; tmin = tmin_; tmax = tmax_; cur = 0; sp = 0; hit = -1; done = false; }
	__device__ __forceinline__ bool heavy() const { return cur < 0; }                 // next step is a leaf (primitive tests), not a node
	template <bool kAnyHit> __device__ __forceinline__ void step(const SceneView& sc, const WalkStack& stack)
	{
		if (cur >= 0)
		{
			const float4* __restrict__ nodes = sc.nodes;
			const float4 n0 = nodes[4 * cur + 0], n1 = nodes[4 * cur + 1], n2 = nodes[4 * cur + 2], n3 = nodes[4 * cur + 3];
			const float lx0 = (n0.x - o.x) * ix, lx1 = (n0.w - o.x) * ix;
			const float ly0 = (n0.y - o.y) * iy, ly1 = (n1.x - o.y) * iy;
			const float lz0 = (n0.z - o.z) * iz, lz1 = (n1.y - o.z) * iz;
			const float ln = fmaxf(fmaxf(fminf(lx0, lx1), fminf(ly0, ly1)), fmaxf(fminf(lz0, lz1), tmin));
			const float lf = fminf(fminf(fmaxf(lx0, lx1), fmaxf(ly0, ly1)), fminf(fmaxf(lz0, lz1), tmax));
			const float rx0 = (n1.z - o.x) * ix, rx1 = (n2.y - o.x) * ix;
			const float ry0 = (n1.w - o.y) * iy, ry1 = (n2.z - o.y) * iy;
			const float rz0 = (n2.x - o.z) * iz, rz1 = (n2.w - o.z) * iz;
			const float rn = fmaxf(fmaxf(fminf(rx0, rx1), fminf(ry0, ry1)), fmaxf(fminf(rz0, rz1), tmin));
			const float rf = fminf(fminf(fmaxf(rx0, rx1), fmaxf(ry0, ry1)), fminf(fmaxf(rz0, rz1), tmax));
			const bool hl = ln <= lf * 1.000002f, hr = rn <= rf * 1.000002f;
			const int cl = __float_as_int(n3.x), cr = __float_as_int(n3.y);
			if (hl && hr)
			{
				const bool leftFirst = kAnyHit ? true : (ln <= rn);
				cur = leftFirst ? cl : cr;
				stack.put(sp, leftFirst ? cr : cl); sp++;
			}
			else if (hl) cur = cl;
			else if (hr) cur = cr;
			else if (sp > 0) { sp--; cur = stack.get(sp); }
			else done = true;
		}
		else
		{   // one leaf per step (measured: one primitive per step is better without the vote, 28.3 -> 23.9 ms, worse with it, 21.9 -> 23.2)
			const int e = -cur - 1, first = e >> 4, count = (e & 15) + 1;
			for (int k = 0; k < count; k++)
				if (prim_hit<4>(sc.prims, first + k, o, d, tmin, tmax)) { hit = first + k; if (kAnyHit) { done = true; return; } }
			if (sp > 0) { sp--; cur = stack.get(sp); } else done = true;
		}
	}
};

template <> struct Walker<5>
{
	V3 o, d, rd; float tmin, tmax; int cur, sp, hit, lfirst, lcount, lk; bool done;
	__device__ __forceinline__ void start(V3 o_, V3 d_, float tmin_, float tmax_)
	{ o = o_; d = d_; rd = mk(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z)); tmin = tmin_; tmax = tmax_; cur = 0; sp = 0; hit = -1; lcount = 0; lk = 0; lfirst = 0; done = false; }
	__device__ __forceinline__ bool heavy() const { return lcount > 0; }             // next step tests an object of the leaf just entered
	template <bool kAnyHit> __device__ __forceinline__ void step(const SceneView& sc, const WalkStack& stack)
	{   // traverse_ref: the node's own box, then left, then right; every object of a leaf in order
		if (lcount > 0)
		{
			if (prim_hit<4>(sc.prims, lfirst + lk, o, d, tmin, tmax)) { hit = lfirst + lk; if (kAnyHit) { done = true; return; } }
			if (++lk < lcount) return;
			lcount = 0; lk = 0;
		}
		else
		{
			const float4* __restrict__ nodes = sc.nodes;
			const float4 n0 = nodes[2 * cur], n1 = nodes[2 * cur + 1];
			const bool ok = ref_box(n0, n1, o, d, rd, tmin, tmax);
			const int left = __float_as_int(n0.w), right = __float_as_int(n1.w);
			if (ok && left >= 0) { stack.put(sp, right); sp++; cur = left; return; }
			if (ok && right > 0) { lfirst = -left - 1; lcount = right; lk = 0; return; }
		}
		if (sp == 0) { done = true; return; }
		sp--; cur = stack.get(sp);
	}
};

template <> struct Walker<3>
{
	V3 o, d; float ix, iy, iz, tmin, tmax; unsigned int ngx, ngy, tgx, tgy, octinv; int sp, hit; bool done;
	__device__ __forceinline__ void start(V3 o_, V3 d_, float tmin_, float tmax_)
	{
		o = o_; d = d_; ix = 1.0f / d.x; iy = 1.0f / d.y; iz = 1.0f / d.z; tmin = tmin_; tmax = tmax_;
		octinv = 7u - ((d.x < 0 ? 1u : 0u) | (d.y < 0 ? 2u : 0u) | (d.z < 0 ? 4u : 0u));
		ngx = 0; ngy = 0x80000000u; tgx = tgy = 0; sp = 0; hit = -1; done = false;       // the root: group base 0, one pending inner hit
	}
	__device__ __forceinline__ bool heavy() const { return tgy == 0u && ngy > 0x00ffffffu; }   // next step decodes a wide node (8 boxes), not a primitive / pop
	template <bool kAnyHit> __device__ __forceinline__ void step(const SceneView& sc, const WalkStack& stack)
	{
		if (tgy)
		{   // one primitive
			const int j = __ffs((int)tgy) - 1;
			tgy &= tgy - 1;
			if (prim_hit<4>(sc.prims, (int)tgx + j, o, d, tmin, tmax)) { hit = (int)tgx + j; if (kAnyHit) done = true; }
		}
		else if (ngy > 0x00ffffffu)
		{   // one wide node (traverse_wide)
			const uint4* __restrict__ wide = sc.wide;
			const unsigned int bit = 31u - (unsigned int)__clz((int)ngy);
			ngy &= ~(1u << bit);
			if (ngy > 0x00ffffffu) { stack.put(2 * sp, (int)ngx); stack.put(2 * sp + 1, (int)ngy); sp++; }
			const unsigned int slot = (bit - 24u) ^ octinv;
			const unsigned int rel = (unsigned int)__popc(ngy & 0xffu & ~(0xffffffffu << slot));
			const unsigned int idx = ngx + rel;
			const uint4 q0 = wide[5 * idx + 0], q1 = wide[5 * idx + 1], q2 = wide[5 * idx + 2], q3 = wide[5 * idx + 3], q4 = wide[5 * idx + 4];
			const float sx = __uint_as_float(((q0.w & 0xffu) ) << 23), sy = __uint_as_float(((q0.w >> 8) & 0xffu) << 23), sz = __uint_as_float(((q0.w >> 16) & 0xffu) << 23);
			const float px = __uint_as_float(q0.x) - o.x, py = __uint_as_float(q0.y) - o.y, pz = __uint_as_float(q0.z) - o.z;
			unsigned int hm = 0;
			#pragma unroll
			for (int i = 0; i < 8; i++)
			{
				const int sh = 8 * (i & 3);
				const unsigned int m = ((i < 4 ? q1.z : q1.w) >> sh) & 0xffu;
				const float lx = (float)(((i < 4 ? q2.x : q2.y) >> sh) & 0xffu), ly = (float)(((i < 4 ? q2.z : q2.w) >> sh) & 0xffu), lz = (float)(((i < 4 ? q3.x : q3.y) >> sh) & 0xffu);
				const float hx = (float)(((i < 4 ? q3.z : q3.w) >> sh) & 0xffu), hy = (float)(((i < 4 ? q4.x : q4.y) >> sh) & 0xffu), hz = (float)(((i < 4 ? q4.z : q4.w) >> sh) & 0xffu);
				const float x0 = fmaf(lx, sx, px) * ix, x1 = fmaf(hx, sx, px) * ix;
				const float y0 = fmaf(ly, sy, py) * iy, y1 = fmaf(hy, sy, py) * iy;
				const float z0 = fmaf(lz, sz, pz) * iz, z1 = fmaf(hz, sz, pz) * iz;
				const float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), tmin));
				const float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), tmax));
				if (m != 0u && tn <= tf * 1.000002f)
				{
					const bool inner = (m & 0x18u) == 0x18u;
					const unsigned int bits = inner ? 1u : (m >> 5), at = inner ? 24u + ((m & 7u) ^ octinv) : (m & 31u);
					hm |= bits << at;
				}
			}
			ngx = q1.x; ngy = (hm & 0xff000000u) | (q0.w >> 24);
			tgx = q1.y; tgy = hm & 0x00ffffffu;
		}
		else if (sp > 0) { sp--; ngx = (unsigned int)stack.get(2 * sp); ngy = (unsigned int)stack.get(2 * sp + 1); }
		else done = true;
	}
};

// ---- round 3: 4-wide tree with quantised child boxes for the CLOSEST-HIT rays of large scenes ------------------------------------
// The lane-refill kernels on the 280k-triangle scene are bound by the vector-memory path: every lane fetches its own 64-byte node per
// step, ~19 cycles per 1-KiB wave load instruction with 32 waves per CU (profiles/r02g_c3_pmc_sq.txt: 562 node / leaf steps of ~2,400
// cycles per wave), and a binary node spends those 64 bytes on TWO child boxes.  Here a 64-byte node holds FOUR children:
//   q0 = (p.x, p.y, p.z, e.x | e.y << 8 | e.z << 16 | valid << 24)     node origin, per-axis scale exponents, valid-child mask
//   q1 = (ref0, ref1, ref2, ref3)        child references in the binary tree's encoding: >= 0 node index, < 0 leaf (first << 4 | count - 1)
//   q2 = (lo.x[0..3], lo.y[0..3], lo.z[0..3], hi.x[0..3])   q3 = (hi.y[0..3], hi.z[0..3], -, -)      one byte per child and plane
// child plane = p + q * 2^e, rounded outward by the host until the fp32 value the device computes is conservative (as for the 8-wide
// tree).  Unlike the 8-wide walk (octant order) the hit children are visited in EXACT near-to-far order of their dequantised entry
// distances: each hit child's rank among the hits (six comparisons) says where it goes -- rank 0 is walked next, the others onto the
// stack, farthest first.  Half the node steps per ray of the binary walk for the same bytes per step.  The quantised boxes contain
// the binary tree's boxes, so every hit the binary walk finds is found (never a dropped hit; fringe hits as DESIGN.md "Numerics").
//
// kCert (Walker<6>, reference semantics on large scenes): the CERTIFIED walk.  The 4-wide tree is then built over the LEAVES of the caller's
// (the reference's) tree -- one leaf of this tree = one leaf of that tree, its objects tested in that leaf's order -- and the ordered walk
// is used to find the closest accepted hit among all leaves the reference could visit, together with a proof that the reference finds
// exactly that hit:
//  * the reference reaches a leaf only through FBounds3::Intersect of the leaf's own box (bvh.h:94-103, the node that wraps the leaf), so the
//    leaves that can contribute are those whose box the ray passes; this walk's boxes contain them (quantised outward), and a node is
//    culled by distance only beyond tmax + cert_pad * tmax^2 (a hit accepted in the fp32 fringe just IN FRONT of its leaf's box lies
//    nearer than the box entry; the slack is the part of this scheme that is empirical, DESIGN.md "Certified walk");
//  * for the closest hit found, (t, p): if the box test of geometry.cc:10-30, verbatim, passes on p's leaf box with max_t = t, it passes
//    on that box and on every ancestor's (their planes enclose it; correctly rounded subtraction and division are monotone) for every
//    max_t >= t -- and the reference's max_t never drops below t before it reaches the leaf, because no accepted hit is nearer.  So the
//    reference visits the leaf and accepts p.  `certain()` evaluates exactly that;
//  * two accepted hits at the same distance (the reference keeps whichever it reaches first), a certificate that fails, a ray parallel to
//    an axis: the ray is UNSURE and is walked again the reference's way (Walker<5>) -- a few rays in 10^4.
// Shadow rays: any accepted hit with a passing certificate means the reference's walk ends occluded; none found means visible.
#ifdef JP_WALK_STATS
// diagnostic builds only (tools/walk_stats.py): node steps, leaf steps, primitive tests, stack pushes of the 4-wide walks -- [0..3] closest-hit rays, [4..7] shadow rays
__device__ unsigned long long g_walk_stats[8];
#define JP_WS(i, v) atomicAdd(&g_walk_stats[(kAnyHit ? 4 : 0) + (i)], (unsigned long long)(v))
// per wave iteration of the refill kernels, [0..15] closest-hit, [16..31] shadow: 0 node turns, 1 lanes stepping in them, 2 live lanes at node turns, 3 leaf turns,
// 4 lanes stepping, 5 live lanes, 6 refill iterations, 7 lanes refilled, 8 turns with the pool empty (drain), 9 live lanes in those, 10 lanes stepping in those
__device__ unsigned long long g_turn_stats[32];
#define JP_TURN(i, v) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_turn_stats[(kAnyHit ? 16 : 0) + (i)], (unsigned long long)(v)); } while (0)
#else
#define JP_WS(i, v) do { } while (0)
#define JP_TURN(i, v) do { } while (0)
#endif
template <bool kCert> struct WalkerQ4
{
	V3 o, d; float ix, iy, iz, tmin, tmax; int cur, sp, hit; bool done, unsure, from_eye;
	__device__ __forceinline__ void start(V3 o_, V3 d_, float tmin_, float tmax_)
	{
		// reciprocal direction: the hardware approximation (1 ulp, 8 issue cycles against the 42 of an IEEE division -- three per ray): every slab distance of an axis is scaled by the same
		// 1 +- 1.2e-7, far inside the 2e-6 slack of the box test below (the flat leaf list has always used it: flat_boxes); 0 and denormals give +-inf: the axis-parallel case
		o = o_; d = d_; ix = __builtin_amdgcn_rcpf(d.x); iy = __builtin_amdgcn_rcpf(d.y); iz = __builtin_amdgcn_rcpf(d.z); tmin = tmin_; tmax = tmax_; cur = 0; sp = 0; hit = -1; done = false;
		unsure = kCert && (d.x == 0.f || d.y == 0.f || d.z == 0.f); from_eye = false;
	}
	// kCert: a ray that starts at the camera position is not culled by distance at the children flagged "edge-on to the camera" (jp_upload_scene): the
	// triangles whose plane passes through the eye are the ones a camera ray can lie in to within fp32 noise (see above)
	__device__ __forceinline__ void mark_origin(const SceneView& sc) { from_eye = kCert && o.x == sc.cam.pos[0] && o.y == sc.cam.pos[1] && o.z == sc.cam.pos[2]; }
	__device__ __forceinline__ bool heavy() const { return cur < 0; }                 // next step is a leaf (primitive tests), not a node
	// FBounds3::Intersect (geometry.cc:10-30) on the reference leaf box of primitive p with max_t = t
	__device__ __forceinline__ bool leaf_box_passes(const SceneView& sc, int p, float t) const
	{ const float4 b0 = sc.refbox[2 * (size_t)p], b1 = sc.refbox[2 * (size_t)p + 1]; return ref_box_exact(b0, b1, o, d, tmin, t); }
	// closest hit: is (tmax, hit) what the reference returns?  (any-hit rays are decided as they go: hit >= 0 is certain, unsure says the rest)
	__device__ __forceinline__ bool certain(const SceneView& sc) const { return !unsure && (hit < 0 || leaf_box_passes(sc, hit, tmax)); }
	__device__ __forceinline__ void pop(const WalkStack& stack) { if (sp > 0) { sp--; cur = stack.get(sp); } else done = true; }
	// one interior node: the four quantised child boxes, the hit ones in exact near-to-far order
	template <bool kAnyHit> __device__ __forceinline__ void node_step(const SceneView& sc, const WalkStack& stack)
	{
		const uint4* __restrict__ nd = sc.q4 + 4 * (size_t)cur;
		const uint4 q0 = nd[0], q1 = nd[1], q2 = nd[2]; const uint2 q3 = *(const uint2*)(nd + 3);
		const unsigned int eyebits = kCert ? ((const unsigned int*)(nd + 3))[2] : 0u;
		// slab distance of plane byte q on axis x: ((p.x + q * 2^e.x) - o.x) / d.x = q * A.x + B.x with A = 2^e / d, B = (p - o) / d: one
		// conversion and one fma per plane.  Rounding: |error| <~ 3e-7 * (node extent) / |d| -- covered by the host's padding of every child
		// box by 1e-6 of the node's extent before it is quantised outward (jp_upload_scene), on top of the 2e-6 relative slack below.
		const float ax = __uint_as_float((q0.w & 0xffu) << 23) * ix, ay = __uint_as_float(((q0.w >> 8) & 0xffu) << 23) * iy, az = __uint_as_float(((q0.w >> 16) & 0xffu) << 23) * iz;
		const float bx = (__uint_as_float(q0.x) - o.x) * ix, by = (__uint_as_float(q0.y) - o.y) * iy, bz = (__uint_as_float(q0.z) - o.z) * iz;
		// near / far plane of each axis by the sign of the direction: min / max of the two slab distances without computing both orders
		// (a NaN -- 0 * inf on an axis the ray is parallel to -- is dropped by fmaxf / fminf: that axis then does not constrain, conservative)
		const bool nx = ix >= 0.f, ny = iy >= 0.f, nz = iz >= 0.f;
		const unsigned int axn = nx ? q2.x : q2.w, axf = nx ? q2.w : q2.x;
		const unsigned int ayn = ny ? q2.y : q3.x, ayf = ny ? q3.x : q2.y;
		const unsigned int azn = nz ? q2.z : q3.y, azf = nz ? q3.y : q2.z;
		float tn[4]; bool hc[4]; int nh = 0;
		const float tcull = kCert ? fmaf((from_eye ? sc.cert_pad_eye : sc.cert_pad) * tmax, tmax, tmax) : tmax;
		const float tcull_eye = (kCert && from_eye) ? JP_INF : tcull;                  // for the children with the edge-on flag (bits 0..3 of q3.z)
		#pragma unroll
		for (int i = 0; i < 4; i++)
		{
			const int sh = 8 * i;
			const float x0 = fmaf((float)((axn >> sh) & 0xffu), ax, bx), x1 = fmaf((float)((axf >> sh) & 0xffu), ax, bx);
			const float y0 = fmaf((float)((ayn >> sh) & 0xffu), ay, by), y1 = fmaf((float)((ayf >> sh) & 0xffu), ay, by);
			const float z0 = fmaf((float)((azn >> sh) & 0xffu), az, bz), z1 = fmaf((float)((azf >> sh) & 0xffu), az, bz);
			const float t0 = fmaxf(fmaxf(x0, y0), fmaxf(z0, tmin));
			const float tf = fminf(fminf(x1, y1), fminf(z1, (kCert && ((eyebits >> i) & 1u)) ? tcull_eye : tcull));
			hc[i] = t0 <= tf * 1.000002f;                             // (an unused slot holds the empty box lo = 255 > hi = 0 on every axis and a reference to primitive 0 as a one-primitive leaf: it fails here, and a pass would cost one harmless test -- no valid bit to check)
			tn[i] = hc[i] ? fminf(t0, 3.0e38f) : JP_INF;             // a missed child ranks behind every hit one
			nh += hc[i] ? 1 : 0;
		}
		const int ref[4] = { (int)q1.x, (int)q1.y, (int)q1.z, (int)q1.w };
		// rank of a child = number of children entered before it (ties by child number: a total order; six comparisons); the hit
		// children have the ranks 0 .. nh-1.  Any-hit rays: the children's own order (rank among the hit ones).
		int r[4] = { 0, 0, 0, 0 };
		if (kAnyHit) { r[1] = hc[0] ? 1 : 0; r[2] = r[1] + (hc[1] ? 1 : 0); r[3] = r[2] + (hc[2] ? 1 : 0); }
		else
		{   // s_ij = 1 iff child j is entered before child i.  The entry distances are non-negative floats (>= tmin; a missed child: +inf), which order like their bit patterns, and
			// two such patterns differ by less than 2^31: the sign bit of the integer difference is the comparison -- a subtraction and a shift of the 2.6-cycle class instead of a
			// v_cmp + v_cndmask pair at 4.4 each (profiles/r04a_valu_issue_cost.txt); ties go to the lower child number, as before
			// (written as instructions: from the expression `(unsigned)(bj - bi) >> 31` the compiler proves the operands non-negative and goes back to v_cmp_lt_i32 + v_cndmask)
			auto before = [](float tj, float ti) -> int { int sgn; asm("v_sub_u32 %0, %1, %2\n\tv_lshrrev_b32 %0, 31, %0" : "=&v"(sgn) : "v"(tj), "v"(ti)); return sgn; };
			const int s01 = before(tn[1], tn[0]), s02 = before(tn[2], tn[0]), s03 = before(tn[3], tn[0]), s12 = before(tn[2], tn[1]), s13 = before(tn[3], tn[1]), s23 = before(tn[3], tn[2]);
			r[0] = s01 + s02 + s03; r[1] = 1 - s01 + s12 + s13; r[2] = 2 - s02 - s12 + s23; r[3] = 3 - s03 - s13 - s23;
		}
		// rank 0 is walked next, the others go on the stack, farthest first.  No branch per child: while the three possible entries
		// fit the LDS part of the stack, every child stores -- the ones with nothing to push into the word behind the stack's LDS part
		// (the kernels allocate cap + 1 words per thread)
		if (sp + 3 <= stack.cap)
		{
			#pragma unroll
			for (int i = 0; i < 4; i++)
			{
				const bool push = hc[i] && r[i] != 0;
				stack.lds[(push ? sp + nh - 1 - r[i] : stack.cap) * JP_BLOCK] = ref[i];
				cur = (hc[i] && r[i] == 0) ? ref[i] : cur;
			}
		}
		else
		{
			#pragma unroll
			for (int i = 0; i < 4; i++)
				if (hc[i]) { if (r[i] == 0) cur = ref[i]; else stack.put(sp + nh - 1 - r[i], ref[i]); }
		}
		sp += nh > 0 ? nh - 1 : 0;
		JP_WS(0, 1); JP_WS(3, nh > 0 ? nh - 1 : 0);
#ifdef JP_DBG_VALU_PAD
		{   // diagnostic builds only: JP_DBG_VALU_PAD extra vector instructions per node step, same memory behaviour (profiles/r04a_walker_experiments.txt)
			float pad = tmax;
			#pragma unroll
			for (int i = 0; i < JP_DBG_VALU_PAD; i++) asm volatile("v_add_f32 %0, %0, %0" : "+v"(pad));
			if (pad == 1.2345e-30f) hit = -3;
		}
#endif
		if (nh == 0) pop(stack);
	}
	// one leaf: every primitive in order (Walker<0>: one leaf per step)
	template <bool kAnyHit> __device__ __forceinline__ void leaf_step(const SceneView& sc, const WalkStack& stack)
	{
		const int e = -cur - 1, first = e >> 4, count = (e & 15) + 1;
		JP_WS(1, 1); JP_WS(2, count);
		if (kCert)
		{
			for (int k = 0; k < count; k++)
			{
				// accept distance <= tmax to see ties: prim_hit's `distance < bound` with the next float above tmax as the bound (inf stays inf)
				float tm = __int_as_float(__float_as_int(tmax) + (tmax < JP_INF ? 1 : 0));
				if (!prim_hit<4>(sc.prims, first + k, o, d, tmin, tm)) continue;
				if (kAnyHit)
				{
					if (!(tm < tmax)) continue;                                       // distance == max_t: not a hit for the reference either
					if (leaf_box_passes(sc, first + k, tm)) { hit = first + k; tmax = tm; done = true; return; }
					unsure = true;                                                    // accepted, but the reference may never reach this leaf: go on looking
				}
				else if (tm < tmax) { tmax = tm; hit = first + k; }
				else if (hit >= 0) unsure = true;                                     // a second primitive at exactly the closest distance
			}
		}
		else
		for (int k = 0; k < count; k++)
			if (prim_hit<4>(sc.prims, first + k, o, d, tmin, tmax)) { hit = first + k; if (kAnyHit) { done = true; return; } }
		pop(stack);
	}
	template <bool kAnyHit> __device__ __forceinline__ void step(const SceneView& sc, const WalkStack& stack)
	{ if (cur >= 0) node_step<kAnyHit>(sc, stack); else leaf_step<kAnyHit>(sc, stack); }
};
template <> struct Walker<4> : WalkerQ4<false> {};
template <> struct Walker<6> : WalkerQ4<true> {};

// a walker driven to the end by one lane (k_trace, k_other): the same steps as under lane refill
template <int kMode, bool kAnyHit>
__device__ __forceinline__ int walk_ray(const SceneView& sc, V3 o, V3 d, float tmin, float& tmax, const WalkStack& stack)
{
	if constexpr (kMode == 6)
	{   // certified walk; an unsure ray is walked again the reference's way
		Walker<6> w; w.start(o, d, tmin, tmax); w.mark_origin(sc);
		while (!w.done) w.template step<kAnyHit>(sc, stack);
		if (kAnyHit ? (w.hit >= 0 || !w.unsure) : w.certain(sc)) { tmax = w.tmax; return w.hit; }
		return walk_ray<5, kAnyHit>(sc, o, d, tmin, tmax, stack);
	}
	else
	{
		Walker<kMode> w; w.start(o, d, tmin, tmax);
		while (!w.done) w.template step<kAnyHit>(sc, stack);
		tmax = w.tmax;
		return w.hit;
	}
}

} // namespace jp
