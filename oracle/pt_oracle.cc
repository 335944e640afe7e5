// oracle/pt_oracle.cc -- TEST INFRASTRUCTURE ONLY.  Never linked, imported or called by the shipped product
// (jet-pbrt_amd/): only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
//
// A plain-C++ CPU restatement of the reference hot path over the flat JpScene arrays of
// include/jetpbrt_amd.h.  Every function cites the reference lines it follows (paths under
// /root/reference/src).  Arithmetic is fp32 with the reference's exact operation order; build with
// g++ -O2 for baseline x86-64 only (no -march=native, no -ffast-math: no FMA contraction), as the
// reference build in oracle/ref_build does.
//
// PINNING: this restatement is pinned against the UNMODIFIED reference compiled in this container
// (oracle/ref_build -> oracle/_ref/libjp_ref.so): whole films are bit-identical for the stock
// mt19937_64 stream (tier T0) and for the counter stream (tier T1), see tests/test_oracle_vs_reference.py,
// and against the golden films/vectors generated from that library and committed under tests/golden/.
#include "jetpbrt_amd.h"
#include "jp_counter_rng.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <thread>
#include <atomic>
#include <limits>

namespace {

// ---------------------------------------------------------------------------------------------
// vectors / colours (geometry.h:65-159, color.h:13-69): component-wise fp32, left-to-right sums
// ---------------------------------------------------------------------------------------------
struct V3 { float x, y, z; };
inline V3 mk(float x, float y, float z) { V3 v = { x, y, z }; return v; }
inline V3 ld3(const float* p) { return mk(p[0], p[1], p[2]); }
inline V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
inline V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(float s, V3 a) { return mk(a.x * s, a.y * s, a.z * s); }
inline V3 operator/(V3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
inline V3 cmul(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }      // FColor * FColor
inline V3 cdiv(V3 a, V3 b) { return mk(a.x / b.x, a.y / b.y, a.z / b.z); }      // FColor / FColor
inline V3 splat(float v) { return mk(v, v, v); }                                // FColor(Float)
inline V3 csqrt(V3 a) { return mk(std::sqrt(a.x), std::sqrt(a.y), std::sqrt(a.z)); }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }      // geometry.h:107
inline float len2(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }           // geometry.h:101
inline float len(V3 a) { return std::sqrt(len2(a)); }                           // geometry.h:102
inline V3 normalize(V3 a) { return a / len(a); }                                // geometry.h:104 (3 divides)
inline V3 cross(V3 a, V3 v) { return mk(a.y * v.z - a.z * v.y, a.z * v.x - a.x * v.z, a.x * v.y - a.y * v.x); } // geometry.h:108-119
inline float absdot(V3 a, V3 b) { return std::abs(dot(a, b)); }
inline bool isblack(V3 c) { return c.x == 0.f && c.y == 0.f && c.z == 0.f; }   // color.h:50
inline float maxcomp(V3 c) { float m = (c.y < c.z) ? c.z : c.y; return (c.x < m) ? m : c.x; }   // color.h:40-43 (std::max)
inline float comp(V3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }

// std::min / std::max exactly as libstdc++ defines them (NaN-sensitive; geometry.cc:19-26)
inline float smin(float a, float b) { return (b < a) ? b : a; }
inline float smax(float a, float b) { return (a < b) ? b : a; }
// pbrt.h:73-83
inline float clampf(float v, float lo, float hi) { if (v < lo) return lo; else if (v > hi) return hi; else return v; }

const float kPi = (float)3.14159265358979323846;   // pbrt.h:39-46
const float k2Pi = 2.0f * kPi;
const float kPiOver2 = kPi / 2.0f;
const float kPiOver4 = kPi / 4.0f;
const float kInvPi = 1.0f / kPi;
const float kInf = std::numeric_limits<float>::infinity();

struct Ray { V3 o, d; float tmin; mutable float tmax; };              // geometry.h:384-420
inline Ray mkray(V3 o, V3 d, float t0 = 0.001f, float t1 = kInf) { Ray r; r.o = o; r.d = d; r.tmin = t0; r.tmax = t1; return r; }

struct Isect { V3 p, n, wo; int prim; };                              // shape.h:33-77

// ---------------------------------------------------------------------------------------------
// samplers (sampler.h:16-54, 64-105, 130-156)
// ---------------------------------------------------------------------------------------------
struct Mt64                                                          // std::mt19937_64 (ISO C++ [rand.predef])
{
	uint64_t mt[312]; int idx;
	explicit Mt64(uint64_t seed) { mt[0] = seed; for (int i = 1; i < 312; i++) mt[i] = 6364136223846793005ULL * (mt[i - 1] ^ (mt[i - 1] >> 62)) + (uint64_t)i; idx = 312; }
	uint64_t next()
	{
		if (idx >= 312)
		{
			for (int i = 0; i < 312; i++)
			{
				uint64_t x = (mt[i] & 0xFFFFFFFF80000000ULL) | (mt[(i + 1) % 312] & 0x7FFFFFFFULL);
				uint64_t xa = x >> 1; if (x & 1ULL) xa ^= 0xB5026F5AA96619E9ULL;
				mt[i] = mt[(i + 156) % 312] ^ xa;
			}
			idx = 0;
		}
		uint64_t y = mt[idx++];
		y ^= (y >> 29) & 0x5555555555555555ULL; y ^= (y << 17) & 0x71D67FFFEDA60000ULL;
		y ^= (y << 37) & 0xFFF7EEE000000000ULL; y ^= y >> 43;
		return y;
	}
};

struct Sampler
{
	int mode; uint32_t seed; int spp;
	Mt64 mt; uint32_t key, dim; int sample_index;
	Sampler(int mode, uint32_t seed, int spp) : mode(mode), seed(seed), spp(spp), mt(1234), key(0), dim(0), sample_index(0) {}   // sampler.h:26
	// libstdc++ uniform_real_distribution<float>(0,1) over mt19937_64: one engine draw, float(u64) / 2^64,
	// clamped below 1 (std::generate_canonical); SURVEY.md section 8c "RNG facts pinned by probe"
	float stock() { float r = (float)mt.next() / 18446744073709551616.0f; if (r >= 1.0f) r = std::nextafter(1.0f, 0.0f); return r; }
	float get1() { if (mode == JP_SAMPLER_DEBUG) return 0.5f; return mode == JP_SAMPLER_STOCK_MT19937 ? stock() : jp_rng_float(key, dim++); }   // FDebugSampler sampler.h:118                  // sampler.h:140-142
	void get2(float& x, float& y)                                                                                 // sampler.h:144-146, 49-52
	{
		if (mode == JP_SAMPLER_DEBUG) { x = y = 0.5f; return; }                // FDebugSampler sampler.h:119
		if (mode == JP_SAMPLER_STOCK_MT19937) { y = stock(); x = stock(); }   // g++ evaluates the 2nd ctor argument first
		else { x = jp_rng_float(key, dim++); y = jp_rng_float(key, dim++); }
	}
	void start_pixel() { sample_index = 0; }                                                                      // sampler.h:86-89
	bool next_sample() { sample_index++; return sample_index < spp; }                                             // sampler.h:91-95
	void camera_sample(float px, float py, float& fx, float& fy)                                                  // sampler.h:148-155
	{
		if (mode != JP_SAMPLER_STOCK_MT19937) { key = jp_rng_key(seed, (uint32_t)(int)px, (uint32_t)(int)py, (uint32_t)sample_index); dim = 0; }
		float ux, uy; get2(ux, uy); fx = px + ux; fy = py + uy;
	}
};

// sampler with scripted values for known-answer vectors
struct Script { const float* v; int n; int pos; float next() { float r = pos < n ? v[pos] : 0.5f; pos++; return r; } };

// A small indirection so Li() can draw from either a Sampler or a Script.
struct Draw
{
	Sampler* s; Script* k;
	float get1() { return s ? s->get1() : k->next(); }
	void get2(float& x, float& y) { if (s) s->get2(x, y); else { x = k->next(); y = k->next(); } }
};

// ---------------------------------------------------------------------------------------------
// shapes (shape.h)
// ---------------------------------------------------------------------------------------------
struct Tri { V3 p0, p1, p2, n; };
struct Rect { V3 p0, p1, p2, p3, n; };
struct Frame { V3 s, t, n; };
inline Frame frame_from_z(V3 nn)                                         // geometry.h:345-349, 371-376
{
	Frame f; f.n = normalize(nn);
	V3 tmp = (std::abs(f.n.x) > 0.99f) ? mk(0, 1, 0) : mk(1, 0, 0);
	f.t = normalize(cross(f.n, tmp));
	f.s = normalize(cross(f.t, f.n));
	return f;
}
inline V3 to_local(const Frame& f, V3 w) { return mk(dot(f.s, w), dot(f.t, w), dot(f.n, w)); }       // geometry.h:352-358
inline V3 to_world(const Frame& f, V3 l) { return f.s * l.x + f.t * l.y + f.n * l.z; }               // geometry.h:360-366
struct Disk { V3 c, n; float r; };                                       // FDisk shape.h:189-275 (n normalised by the ctor)
struct Sph { V3 c; float r; };

// FTriangle::Intersect shape.h:291-327
inline bool tri_intersect(const Tri& T, const Ray& ray, Isect& is)
{
	const V3 oa = T.p0 - ray.o, ob = T.p1 - ray.o, oc = T.p2 - ray.o;
	const V3 v0 = cross(oc, ob), v1 = cross(ob, oa), v2 = cross(oa, oc);
	const float v0d = dot(v0, ray.d), v1d = dot(v1, ray.d), v2d = dot(v2, ray.d);
	if (((v0d < 0) && (v1d < 0) && (v2d < 0)) || ((v0d >= 0) && (v1d >= 0) && (v2d >= 0)))
	{
		const float distance = dot(T.n, oa) / dot(T.n, ray.d);
		if ((distance > ray.tmin) && (distance < ray.tmax))
		{
			ray.tmax = distance;
			is.p = ray.o + distance * ray.d; is.n = T.n; is.wo = -ray.d;
			return true;
		}
	}
	return false;
}

// FRectangle::Intersect shape.h:399-435 (normal flipped toward the ray, :427)
inline bool rect_intersect(const Rect& R, const Ray& ray, Isect& is)
{
	const V3 oa = R.p0 - ray.o, ob = R.p1 - ray.o, oc = R.p2 - ray.o, od = R.p3 - ray.o;
	const V3 v0 = cross(oc, ob), v1 = cross(ob, oa), v2 = cross(oa, od), v3 = cross(od, oc);
	const float v0d = dot(v0, ray.d), v1d = dot(v1, ray.d), v2d = dot(v2, ray.d), v3d = dot(v3, ray.d);
	if (((v0d < 0) && (v1d < 0) && (v2d < 0) && (v3d < 0)) || ((v0d >= 0) && (v1d >= 0) && (v2d >= 0) && (v3d >= 0)))
	{
		const float distance = dot(R.n, oa) / dot(R.n, ray.d);
		if ((distance > ray.tmin) && (distance < ray.tmax))
		{
			ray.tmax = distance;
			is.p = ray.o + distance * ray.d;
			is.n = dot(R.n, ray.d) <= 0 ? R.n : -R.n;
			is.wo = -ray.d;
			return true;
		}
	}
	return false;
}

// FSphere::Intersect shape.h:487-526 (unqualified sqrt -> the double overload, result stored to Float)
inline bool sph_intersect(const Sph& S, const Ray& ray, Isect& is)
{
	V3 oc = ray.o - S.c;
	float a = len2(ray.d);
	float half_b = dot(oc, ray.d);
	float c = len2(oc) - S.r * S.r;
	float disc = half_b * half_b - a * c;
	float t_max = ray.tmax, t_min = ray.tmin;
	if (disc > 0.0)
	{
		float root = (float)sqrt((double)disc);
		float time = 0.0f;
		float root1 = (-half_b - root) / a;
		if (root1 < t_max && root1 > t_min) time = root1;
		else
		{
			float root2 = (-half_b + root) / a;
			if (root2 < t_max && root2 > t_min) time = root2;
			else return false;
		}
		ray.tmax = time;
		is.p = ray.o + time * ray.d;
		is.n = normalize(is.p - S.c);
		is.wo = -ray.d;
		return true;
	}
	return false;
}

// FDisk::Intersect shape.h:200-221; isEqual pbrt.h:97-104 with epsilon = numeric_limits<float>::epsilon()
inline bool disk_intersect(const Disk& D, const Ray& ray, Isect& is)
{
	const float dn = dot(ray.d, D.n);
	if (std::abs(dn - 0.f) <= std::numeric_limits<float>::epsilon() * smax(1.f, smax(std::abs(dn), std::abs(0.f)))) return false;
	const V3 op = D.c - ray.o;
	const float distance = dot(D.n, op) / dot(D.n, ray.d);
	if ((distance > ray.tmin) && (distance < ray.tmax))
	{
		const V3 hp = ray.o + distance * ray.d;
		if (len(D.c - hp) <= D.r)
		{
			ray.tmax = distance;
			is.p = hp; is.n = D.n; is.wo = -ray.d;
			return true;
		}
	}
	return false;
}

struct Box { V3 mn, mx; };
inline Box empty_box()                                                  // geometry.h:249-256
{ const float lo = std::numeric_limits<float>::lowest(), hi = std::numeric_limits<float>::max(); Box b = { mk(hi, hi, hi), mk(lo, lo, lo) }; return b; }
inline V3 vmin(V3 a, V3 b) { return mk(smin(a.x, b.x), smin(a.y, b.y), smin(a.z, b.z)); }
inline V3 vmax(V3 a, V3 b) { return mk(smax(a.x, b.x), smax(a.y, b.y), smax(a.z, b.z)); }
inline Box box2(V3 a, V3 b) { Box r = { vmin(a, b), vmax(a, b) }; return r; }                   // geometry.h:260-264
inline Box join(Box a, V3 p) { return box2(vmin(a.mn, p), vmax(a.mx, p)); }                     // geometry.h:279-282
inline Box join(Box a, Box b) { return box2(vmin(a.mn, b.mn), vmax(a.mx, b.mx)); }              // geometry.h:284-287
inline void thin(Box& b, float e = 0.01f)                                                        // geometry.h:299-304
{
	if (b.mn.x == b.mx.x) { b.mn.x -= e; b.mx.x += e; }
	if (b.mn.y == b.mx.y) { b.mn.y -= e; b.mx.y += e; }
	if (b.mn.z == b.mx.z) { b.mn.z -= e; b.mx.z += e; }
}

// FBounds3::Intersect geometry.cc:10-30
// g_watertight (jp_oracle_set_watertight, off by default): a CONSERVATIVE box test instead -- boxes grown by a relative
// 1e-5 + 1e-4 and an interval test with slack that also keeps NaN slabs.  The reference's test rejects a node when
// `tmax <= tmin`, which rounding makes true for rays grazing a box edge: it drops ~3e-4 of the true nearest hits on the
// 280k-triangle scene, which ones depending on its rand()-driven tree.  The watertight variant is "the reference's
// arithmetic without that defect"; the parity tests use it to show that the device film differs from the reference's on
// such meshes by exactly those dropped hits and nothing else.
static bool g_watertight = false;
inline bool box_hit(const Box& b, const Ray& ray)
{
	float tmin = ray.tmin, tmax = ray.tmax;
	if (g_watertight)
	{
		for (int a = 0; a < 3; a++)
		{
			float mn = comp(b.mn, a), mx = comp(b.mx, a);
			float e = smax(std::fabs(mn), std::fabs(mx)) * 1e-5f + 1e-4f;
			float lo = (mn - e - comp(ray.o, a)) / comp(ray.d, a);
			float hi = (mx + e - comp(ray.o, a)) / comp(ray.d, a);
			float t0 = std::fmin(lo, hi), t1 = std::fmax(lo, hi);            // fmin / fmax drop the NaN of 0 * inf
			tmin = std::fmax(t0, tmin);
			tmax = std::fmin(t1, tmax);
		}
		return !(tmin > tmax * 1.00001f + 1e-4f);
	}
	for (int a = 0; a < 3; a++)
	{
		float lo = (comp(b.mn, a) - comp(ray.o, a)) / comp(ray.d, a);
		float hi = (comp(b.mx, a) - comp(ray.o, a)) / comp(ray.d, a);
		float t0 = smin(lo, hi), t1 = smax(lo, hi);
		tmin = smax(t0, tmin);
		tmax = smin(t1, tmax);
		if (tmax <= tmin) return false;
	}
	return true;
}

// ---------------------------------------------------------------------------------------------
// scene view over JpScene + the reference's BVH (bvh.h:54-146), rebuilt here with the reference's
// algorithm (random axis from libc rand(), std::sort on bbox.min[axis], median split, leaves <= 5)
// ---------------------------------------------------------------------------------------------
struct Node { Box box; int left, right; int first, count; bool leaf; };

struct Scene
{
	const JpScene* js;
	std::vector<Tri> tris; std::vector<Rect> rects; std::vector<Sph> sphs; std::vector<Disk> disks;
	std::vector<Box> primBox;
	std::vector<int> order;       // primitive ids in BVH order (the re-sorted shadow_primitives)
	std::vector<Node> nodes; int root;
	std::vector<float> lightInvArea;   // unused cache slot (areas are recomputed per call like the reference)
	std::vector<int> infiniteLights;   // scene.h:101-104

	Box shape_bounds(int prim) const
	{
		int t = js->prim_shape_type[prim], i = js->prim_shape_index[prim];
		if (t == JP_SHAPE_TRIANGLE) { Box b = box2(tris[i].p0, tris[i].p1); b = join(b, tris[i].p2); thin(b); return b; }             // shape.h:342-349
		if (t == JP_SHAPE_RECTANGLE) { Box b = join(join(box2(rects[i].p0, rects[i].p1), rects[i].p2), rects[i].p3); thin(b); return b; } // shape.h:448-454
		if (t == JP_SHAPE_DISK)                                                                                                       // shape.h:238-251
		{
			const Disk& D = disks[i]; Frame fr = frame_from_z(D.n);
			V3 rb = fr.s * D.r, rt = fr.t * D.r;
			Box b = box2(D.c + rb + rt, D.c + rb + rt); b = join(b, D.c + rb - rt); b = join(b, D.c - rb - rt); b = join(b, D.c - rb + rt); thin(b); return b;
		}
		V3 half = mk(sphs[i].r, sphs[i].r, sphs[i].r); return box2(sphs[i].c + half, sphs[i].c - half);                                // shape.h:540-544
	}

	// pbrt.h:106-120 random_int(0,2) over libc rand()
	static int random_axis()
	{
		float r = rand() * (1.0f / (RAND_MAX + 1.0f));
		float v = 0.0f + (3.0f - 0.0f) * r;
		return (int)v;
	}

	int build(size_t start, size_t end)                                // FBVH_Node ctor bvh.h:59-91
	{
		int axis = random_axis();
		size_t span = end - start;
		int me = (int)nodes.size(); nodes.push_back(Node());
		Node nd; nd.leaf = false; nd.first = nd.count = 0; nd.left = nd.right = -1;
		Box bl = empty_box(), br = empty_box();
		if (span <= 5)
		{
			int lf = (int)nodes.size(); nodes.push_back(Node());
			Node L; L.leaf = true; L.left = L.right = -1; L.first = (int)start; L.count = (int)span;
			Box bb = empty_box();
			for (size_t i = start; i < end; i++) { bb.mn = vmin(bb.mn, primBox[order[i]].mn); bb.mx = vmax(bb.mx, primBox[order[i]].mx); }   // bvh.h:119-129
			L.box = bb; nodes[lf] = L;
			nd.left = lf; bl = bb;
		}
		else
		{
			const std::vector<Box>& pb = primBox;
			std::sort(order.begin() + start, order.begin() + end, [&pb, axis](int a, int b) {            // bvh.h:17-35, 76
				return axis == 0 ? pb[a].mn.x < pb[b].mn.x : (axis == 1 ? pb[a].mn.y < pb[b].mn.y : pb[a].mn.z < pb[b].mn.z); });
			size_t mid = start + span / 2;
			nd.left = build(start, mid); nd.right = build(mid, end);
			bl = nodes[nd.left].box; br = nodes[nd.right].box;
		}
		nd.box = join(bl, br);                                          // bvh.h:86-91
		nodes[me] = nd;
		return me;
	}

	explicit Scene(const JpScene* s) : js(s), root(-1)
	{
		for (int i = 0; i < s->n_triangles; i++) { Tri t = { ld3(s->tri_p0 + 3 * i), ld3(s->tri_p1 + 3 * i), ld3(s->tri_p2 + 3 * i), ld3(s->tri_n + 3 * i) }; tris.push_back(t); }
		for (int i = 0; i < s->n_rectangles; i++) { Rect r = { ld3(s->rect_p0 + 3 * i), ld3(s->rect_p1 + 3 * i), ld3(s->rect_p2 + 3 * i), ld3(s->rect_p3 + 3 * i), ld3(s->rect_n + 3 * i) }; rects.push_back(r); }
		for (int i = 0; i < s->n_spheres; i++) { Sph q = { ld3(s->sph_center + 3 * i), s->sph_radius[i] }; sphs.push_back(q); }
		for (int i = 0; i < s->n_disks; i++) { Disk q = { ld3(s->disk_center + 3 * i), ld3(s->disk_normal + 3 * i), s->disk_radius[i] }; disks.push_back(q); }
		for (int i = 0; i < s->n_primitives; i++) { primBox.push_back(shape_bounds(i)); order.push_back(i); }
		for (int i = 0; i < s->n_lights; i++) if (s->light_type[i] == JP_LIGHT_ENVIRONMENT) infiniteLights.push_back(i);
		if (s->n_primitives > 0) root = build(0, (size_t)s->n_primitives);                                  // scene.cc:20-22
	}

	bool prim_intersect(int prim, const Ray& ray, Isect& is) const     // FPrimitive::Intersect primitive.h:39-48
	{
		int t = js->prim_shape_type[prim], i = js->prim_shape_index[prim];
		bool hit = t == JP_SHAPE_TRIANGLE ? tri_intersect(tris[i], ray, is) : (t == JP_SHAPE_RECTANGLE ? rect_intersect(rects[i], ray, is)
		           : (t == JP_SHAPE_DISK ? disk_intersect(disks[i], ray, is) : sph_intersect(sphs[i], ray, is)));
		if (hit) is.prim = prim;
		return hit;
	}

	bool node_intersect(int n, const Ray& ray, Isect& is) const
	{
		const Node& nd = nodes[n];
		if (nd.leaf)                                                   // FBVH_NodeLeaf::Intersect bvh.h:132-142
		{
			bool hit = false;
			for (int i = 0; i < nd.count; i++) hit |= prim_intersect(order[nd.first + i], ray, is);
			return hit;
		}
		if (!box_hit(nd.box, ray)) return false;                       // FBVH_Node::Intersect bvh.h:94-103 (both children, no ordering)
		bool hl = node_intersect(nd.left, ray, is);
		bool hr = nd.right >= 0 ? node_intersect(nd.right, ray, is) : false;
		return hl || hr;
	}

	bool intersect(const Ray& ray, Isect& is) const { return root >= 0 ? node_intersect(root, ray, is) : false; }   // scene.cc:25-33
};

// ---------------------------------------------------------------------------------------------
// sampling warps (sampling.h)
// ---------------------------------------------------------------------------------------------
inline void concentric_disk(float ux, float uy, float& px, float& py)     // sampling.h:25-50
{
	ux = ux * 2.f - 1.f; uy = uy * 2.f - 1.f;
	if (ux == 0 && uy == 0) { px = 0; py = 0; return; }
	float radius, theta;
	if (std::abs(ux) > std::abs(uy)) { radius = ux; theta = kPiOver4 * (uy / ux); }
	else { radius = uy; theta = kPiOver2 - kPiOver4 * (ux / uy); }
	px = std::cos(theta) * radius; py = std::sin(theta) * radius;
}
inline V3 cosine_hemisphere(float ux, float uy)                          // sampling.h:53-59
{
	float px, py; concentric_disk(ux, uy, px, py);
	float z = std::sqrt(smax(0.f, 1 - px * px - py * py));
	return mk(px, py, z);
}
inline V3 uniform_sphere(float ux, float uy)                             // sampling.h:80-87
{
	float z = 1 - 2 * ux;
	float radius = std::sqrt(smax(0.f, 1.f - z * z));
	float phi = 2 * kPi * uy;
	return mk(radius * std::cos(phi), radius * std::sin(phi), z);
}

// ---------------------------------------------------------------------------------------------
// frame + BSDFs (geometry.h:326-378, bsdf.h, bsdf.cc, microfacet.cc, material.h/.cc)
// ---------------------------------------------------------------------------------------------

enum { BS_REFLECTION = 1, BS_TRANSMISSION = 2, BS_SPECULAR = 4, BS_DIFFUSE = 8, BS_GLOSSY = 16 };   // bsdf.h:208-219
enum { CL_LAMBERT, CL_MIRROR, CL_FRESNEL_SPECULAR, CL_MICROFACET };
enum { FR_CONDUCTOR, FR_DIELECTRIC };

struct Closure
{
	int kind; Frame frame;
	V3 c0, c1;                 // lambert: albedo; mirror: reflectance; glass: Kr,Kt; microfacet: R
	float eta_i, eta_t;        // glass
	float ax, ay;              // TrowbridgeReitz alphas
	int fresnel; V3 feta, fk;  // conductor eta/k (etaI = 1); dielectric (1.5, 1)
	bool delta() const { return kind == CL_MIRROR || kind == CL_FRESNEL_SPECULAR; }
};

struct BsdfSample { V3 f, wi; float pdf; int flags; };
inline BsdfSample empty_sample() { BsdfSample s; s.f = splat(0); s.wi = mk(0, 0, 1); s.pdf = 0; s.flags = 0; return s; }   // bsdf.h:252-265

inline bool same_hemi(V3 a, V3 b) { return a.z * b.z > 0; }                                  // bsdf.h:21
inline float sin2t(V3 w) { return smax(0.f, 1.f - w.z * w.z); }                              // bsdf.h:28-30
inline float sint(V3 w) { return std::sqrt(sin2t(w)); }
inline float tant(V3 w) { return sint(w) / w.z; }
inline float tan2t(V3 w) { return sin2t(w) / (w.z * w.z); }
inline float cosphi(V3 w) { float s = sint(w); return (s == 0) ? 1 : clampf(w.x / s, -1.f, 1.f); }   // bsdf.h:40-43
inline float sinphi(V3 w) { float s = sint(w); return (s == 0) ? 0 : clampf(w.y / s, -1.f, 1.f); }   // bsdf.h:45-48

// fresnel_dielectric bsdf.h:91-122
inline float fresnel_dielectric(float cos_i, float eta_i, float eta_t)
{
	cos_i = clampf(cos_i, -1.f, 1.f);
	bool entering = cos_i > 0.f;
	if (!entering) { float t = eta_i; eta_i = eta_t; eta_t = t; cos_i = std::abs(cos_i); }
	float sin_i = std::sqrt(smax(0.f, 1 - cos_i * cos_i));
	float sin_t = eta_i / eta_t * sin_i;
	if (sin_t >= 1) return 1;
	float cos_t = std::sqrt(smax(0.f, 1 - sin_t * sin_t));
	float r_para = ((eta_t * cos_i) - (eta_i * cos_t)) / ((eta_t * cos_i) + (eta_i * cos_t));
	float r_perp = ((eta_i * cos_i) - (eta_t * cos_t)) / ((eta_i * cos_i) + (eta_t * cos_t));
	return (r_para * r_para + r_perp * r_perp) / 2;
}

// fresnel_conductor bsdf.h:174-197
inline V3 fresnel_conductor(float cosI, V3 etai, V3 etat, V3 k)
{
	cosI = clampf(cosI, -1.f, 1.f);
	V3 eta = cdiv(etat, etai), etak = cdiv(k, etai);
	float cos2 = cosI * cosI, sin2 = 1 - cos2;
	V3 eta2 = cmul(eta, eta), etak2 = cmul(etak, etak);
	V3 t0 = eta2 - etak2 - splat(sin2);
	V3 a2plusb2 = csqrt(cmul(t0, t0) + cmul(eta2 * 4.f, etak2));
	V3 t1 = a2plusb2 + splat(cos2);
	V3 a = csqrt((a2plusb2 + t0) * 0.5f);
	V3 t2 = a * ((float)2 * cosI);
	V3 Rs = cdiv(t1 - t2, t1 + t2);
	V3 t3 = a2plusb2 * cos2 + splat(sin2 * sin2);
	V3 t4 = t2 * sin2;
	V3 Rp = cdiv(cmul(Rs, t3 - t4), t3 + t4);
	return (Rp + Rs) * 0.5f;
}

inline V3 fresnel_eval(const Closure& c, float cosI)                       // bsdf.cc:15-24
{
	if (c.fresnel == FR_CONDUCTOR) return fresnel_conductor(std::abs(cosI), splat(1.f), c.feta, c.fk);
	return splat(fresnel_dielectric(cosI, 1.5f, 1.f));                    // material.cc:21
}

// TrowbridgeReitzDistribution (microfacet.cc:181-189, 202-210, 256-357; microfacet.h:22-30; microfacet.cc:359-365)
inline float tr_D(const Closure& c, V3 wh)
{
	float t2 = tan2t(wh);
	if (std::isinf(t2)) return 0.;
	const float cos4 = (wh.z * wh.z) * (wh.z * wh.z);
	float e = (cosphi(wh) * cosphi(wh) / (c.ax * c.ax) + sinphi(wh) * sinphi(wh) / (c.ay * c.ay)) * t2;
	return 1 / (kPi * c.ax * c.ay * cos4 * (1 + e) * (1 + e));
}
inline float tr_Lambda(const Closure& c, V3 w)
{
	float absTan = std::abs(tant(w));
	if (std::isinf(absTan)) return 0.;
	float alpha = std::sqrt(cosphi(w) * cosphi(w) * c.ax * c.ax + sinphi(w) * sinphi(w) * c.ay * c.ay);
	float a2t2 = (alpha * absTan) * (alpha * absTan);
	return (-1 + std::sqrt(1.f + a2t2)) / 2;
}
inline float tr_G1(const Closure& c, V3 w) { return 1 / (1 + tr_Lambda(c, w)); }
inline float tr_G(const Closure& c, V3 wo, V3 wi) { return 1 / (1 + tr_Lambda(c, wo) + tr_Lambda(c, wi)); }
inline float tr_Pdf(const Closure& c, V3 wo, V3 wh) { return tr_D(c, wh) * tr_G1(c, wo) * absdot(wo, wh) / std::abs(wo.z); }   // sampleVisibleArea = true

// microfacet.cc:256-301.  NOTE the double-precision spots: `cosTheta > .9999`, unqualified sqrt/cos/sin
// (C library double overloads), `tmp > 1e10`.
inline void tr_sample11(float cosTheta, float U1, float U2, float* slope_x, float* slope_y)
{
	if ((double)cosTheta > .9999)
	{
		float r = (float)sqrt((double)(U1 / (1 - U1)));
		float phi = 6.28318530718f * U2;
		*slope_x = (float)((double)r * cos((double)phi));
		*slope_y = (float)((double)r * sin((double)phi));
		return;
	}
	float sinTheta = std::sqrt(smax(0.f, 1.f - cosTheta * cosTheta));
	float tanTheta = sinTheta / cosTheta;
	float a = 1 / tanTheta;
	float G1 = 2 / (1 + std::sqrt(1.f + 1.f / (a * a)));
	float A = 2 * U1 / G1 - 1;
	float tmp = 1.f / (A * A - 1.f);
	if ((double)tmp > 1e10) tmp = (float)1e10;
	float B = tanTheta;
	float D = std::sqrt(smax((float)(B * B * tmp * tmp - (A * A - B * B) * tmp), 0.f));
	float sx1 = B * tmp - D, sx2 = B * tmp + D;
	*slope_x = (A < 0 || sx2 > 1.f / tanTheta) ? sx1 : sx2;
	float S;
	if (U2 > 0.5f) { S = 1.f; U2 = 2.f * (U2 - .5f); }
	else { S = -1.f; U2 = 2.f * (.5f - U2); }
	float z = (U2 * (U2 * (U2 * 0.27385f - 0.73369f) + 0.46341f)) / (U2 * (U2 * (U2 * 0.093073f + 0.309420f) - 1.000000f) + 0.597999f);
	*slope_y = S * z * std::sqrt(1.f + *slope_x * *slope_x);
}
inline V3 tr_sample(V3 wi, float ax, float ay, float U1, float U2)         // microfacet.cc:303-324
{
	V3 ws = normalize(mk(ax * wi.x, ay * wi.y, wi.z));
	float sx, sy; tr_sample11(ws.z, U1, U2, &sx, &sy);
	float tmp = cosphi(ws) * sx - sinphi(ws) * sy;
	sy = sinphi(ws) * sx + cosphi(ws) * sy;
	sx = tmp;
	sx = ax * sx; sy = ay * sy;
	return normalize(mk(-sx, -sy, 1.f));
}
inline V3 tr_sample_wh(const Closure& c, V3 wo, float u0, float u1)        // microfacet.cc:326-357 (visible-area branch)
{
	bool flip = wo.z < 0;
	V3 wh = tr_sample(flip ? -wo : wo, c.ax, c.ay, u0, u1);
	if (flip) wh = -wh;
	return wh;
}

inline V3 reflect(V3 wo, V3 n) { return -wo + 2 * dot(wo, n) * n; }       // bsdf.h:62-67
inline bool refract(V3 wi, V3 n, float eta, V3* wt)                        // bsdf.h:70-88
{
	float cos_i = dot(n, wi);
	float sin2_i = smax(0.f, (float)(1 - cos_i * cos_i));
	float sin2_t = eta * eta * sin2_i;
	if (sin2_t >= 1) return false;
	float cos_t = std::sqrt(1 - sin2_t);
	*wt = eta * -wi + (eta * cos_i - cos_t) * n;
	return true;
}

// local-frame evaluation
inline V3 eval_local(const Closure& c, V3 wo, V3 wi)
{
	switch (c.kind)
	{
	case CL_LAMBERT:                                                       // bsdf.h:347-355
		if (!same_hemi(wo, wi)) return splat(0);
		return c.c0 * kInvPi;
	case CL_MICROFACET:                                                    // bsdf.cc:35-51
	{
		float cosO = std::abs(wo.z), cosI = std::abs(wi.z);
		V3 wh = wi + wo;
		if (cosI == 0 || cosO == 0) return splat(0);
		if (wh.x == 0 && wh.y == 0 && wh.z == 0) return splat(0);
		wh = normalize(wh);
		V3 ff = (dot(wh, mk(0, 0, 1)) < 0) ? -wh : wh;                     // face_forward bsdf.h:23-26
		V3 F = fresnel_eval(c, dot(wi, ff));
		return cmul(c.c0 * tr_D(c, wh) * tr_G(c, wo, wi), F) / (4 * cosI * cosO);
	}
	default: return splat(0);                                              // delta BSDFs bsdf.h:405-408, 468-471
	}
}

inline BsdfSample sample_local(const Closure& c, V3 wo, float ux, float uy)
{
	BsdfSample s = empty_sample();
	switch (c.kind)
	{
	case CL_LAMBERT:                                                       // bsdf.h:362-377
	{
		s.wi = cosine_hemisphere(ux, uy);
		if (wo.z < 0) s.wi.z *= -1;
		s.f = eval_local(c, wo, s.wi);
		s.pdf = same_hemi(wo, s.wi) ? std::abs(s.wi.z) * kInvPi : 0;
		s.flags = BS_REFLECTION | BS_DIFFUSE;
		return s;
	}
	case CL_MIRROR:                                                        // bsdf.h:415-429
		s.wi = mk(-wo.x, -wo.y, wo.z);
		s.f = c.c0 / std::abs(s.wi.z);
		s.pdf = 1;
		s.flags = BS_REFLECTION | BS_SPECULAR;
		return s;
	case CL_FRESNEL_SPECULAR:                                              // bsdf.h:478-539
	{
		if (wo.z == 0.f) return s;
		float F = fresnel_dielectric(wo.z, c.eta_i, c.eta_t);
		if (ux < F)
		{
			s.wi = mk(-wo.x, -wo.y, wo.z);
			s.pdf = F;
			s.f = (c.c0 * F) / std::abs(s.wi.z);
			s.flags = BS_REFLECTION | BS_SPECULAR;
		}
		else
		{
			V3 n = mk(0, 0, 1);
			bool entering = wo.z > 0;
			V3 won = entering ? n : -n;
			float etaI = entering ? c.eta_i : c.eta_t;
			float etaT = entering ? c.eta_t : c.eta_i;
			if (refract(wo, won, etaI / etaT, &s.wi))
			{
				V3 ft = c.c1 * (1 - F);
				ft = ft * ((etaI * etaI) / (etaT * etaT));
				s.pdf = 1 - F;
				s.f = ft / std::abs(s.wi.z);
				s.flags = BS_TRANSMISSION | BS_SPECULAR;
			}
			else s.f = splat(0);
		}
		return s;
	}
	case CL_MICROFACET:                                                    // bsdf.cc:60-78
	{
		if (wo.z == 0) return s;
		V3 wh = tr_sample_wh(c, wo, ux, uy);
		if (dot(wo, wh) < 0) return s;
		V3 wi = reflect(wo, wh);
		if (!same_hemi(wo, wi)) return s;
		s.wi = wi;
		s.f = eval_local(c, wo, wi);
		s.pdf = tr_Pdf(c, wo, wh) / (4 * dot(wo, wh));
		s.flags = BS_REFLECTION | BS_GLOSSY;
		return s;
	}
	}
	return s;
}

inline V3 bsdf_eval(const Closure& c, V3 wo_w, V3 wi_w) { return eval_local(c, to_local(c.frame, wo_w), to_local(c.frame, wi_w)); }   // bsdf.h:284-287
inline BsdfSample bsdf_sample(const Closure& c, V3 wo_w, float ux, float uy)                                                         // bsdf.h:295-301
{ BsdfSample s = sample_local(c, to_local(c.frame, wo_w), ux, uy); s.wi = to_world(c.frame, s.wi); return s; }

// ---------------------------------------------------------------------------------------------
// The rest of the reflection API, by value (JpBsdfDesc): the classes no material instantiates -- FPhongSpecularReflection
// bsdf.h:557-633, BeckmannDistribution microfacet.cc:11-254 (both sampling branches), TrowbridgeReitzDistribution's
// non-visible-area branch microfacet.cc:326-350, FMicrofacetTransmission bsdf.cc:80-145, FresnelNoOp bsdf.h:664-667, general
// FresnelConductor / FresnelDielectric parameters -- next to the ones the materials use.  Plain restatement, libm calls where the
// reference has them (logf / expf / powf / acosf / atanf / tanf / sinf / cosf through the float overloads of <cmath>).
// Pinned bit-exact against the compiled reference by tests/golden/kat_bsdf.npz (ref_driver.cc: ref_bsdf_direct).
// ---------------------------------------------------------------------------------------------
namespace xb
{
const float kInv2Pi = (float)1.0 / ((float)2.0 * kPi);                       // pbrt.h:40,45
struct Dist { int kind; float ax, ay; bool vis; };
inline float cos2phi(V3 w) { return cosphi(w) * cosphi(w); }                 // bsdf.h:50-52
inline float sin2phi(V3 w) { return sinphi(w) * sinphi(w); }
inline Dist make_dist(const JpBsdfDesc& d) { Dist r; r.kind = d.distribution; r.ax = smax(0.001f, d.alpha_x); r.ay = smax(0.001f, d.alpha_y); r.vis = d.sample_visible != 0; return r; }   // microfacet.h:66-69, 82-85

inline float ErfInv(float x)                                                 // microfacet.cc:11-41
{
	float w, p;
	x = clampf(x, -.99999f, .99999f);
	w = -std::log((1 - x) * (1 + x));
	if (w < 5)
	{
		w = w - 2.5f;
		p = 2.81022636e-08f; p = 3.43273939e-07f + p * w; p = -3.5233877e-06f + p * w; p = -4.39150654e-06f + p * w; p = 0.00021858087f + p * w;
		p = -0.00125372503f + p * w; p = -0.00417768164f + p * w; p = 0.246640727f + p * w; p = 1.50140941f + p * w;
	}
	else
	{
		w = std::sqrt(w) - 3;
		p = -0.000200214257f; p = 0.000100950558f + p * w; p = 0.00134934322f + p * w; p = -0.00367342844f + p * w; p = 0.00573950773f + p * w;
		p = -0.0076224613f + p * w; p = 0.00943887047f + p * w; p = 1.00167406f + p * w; p = 2.83297682f + p * w;
	}
	return p * x;
}
inline float Erf(float x)                                                    // microfacet.cc:43-64
{
	float a1 = 0.254829592f, a2 = -0.284496736f, a3 = 1.421413741f, a4 = -1.453152027f, a5 = 1.061405429f, p = 0.3275911f;
	int sign = 1;
	if (x < 0) sign = -1;
	x = std::abs(x);
	float t = 1 / (1 + p * x);
	float y = 1 - (((((a5 * t + a4) * t) + a3) * t + a2) * t + a1) * t * std::exp(-x * x);
	return sign * y;
}
inline void BeckmannSample11(float cosThetaI, float U1, float U2, float* slope_x, float* slope_y)   // microfacet.cc:67-144
{
	if (cosThetaI > .9999f)
	{
		float r = std::sqrt(-std::log(1.0f - U1));
		float sinPhi = std::sin(2 * kPi * U2);
		float cosPhi = std::cos(2 * kPi * U2);
		*slope_x = r * cosPhi; *slope_y = r * sinPhi;
		return;
	}
	float sinThetaI = std::sqrt(smax((float)0, (float)1 - cosThetaI * cosThetaI));
	float tanThetaI = sinThetaI / cosThetaI;
	float cotThetaI = 1 / tanThetaI;
	float a = -1, c = Erf(cotThetaI);
	float sample_x = smax(U1, (float)1e-6f);
	float thetaI = std::acos(cosThetaI);
	float fit = 1 + thetaI * (-0.876f + thetaI * (0.4265f - 0.0594f * thetaI));
	float b = c - (1 + c) * std::pow(1 - sample_x, fit);
	static const float SQRT_PI_INV = 1.f / std::sqrt(kPi);
	float normalization = 1 / (1 + c + SQRT_PI_INV * tanThetaI * std::exp(-cotThetaI * cotThetaI));
	int it = 0;
	while (++it < 10)
	{
		if (!(b >= a && b <= c)) b = 0.5f * (a + c);
		float invErf = ErfInv(b);
		float value = normalization * (1 + b + SQRT_PI_INV * tanThetaI * std::exp(-invErf * invErf)) - sample_x;
		float derivative = normalization * (1 - invErf * tanThetaI);
		if (std::abs(value) < 1e-5f) break;
		if (value > 0) c = b; else a = b;
		b -= value / derivative;
	}
	*slope_x = ErfInv(b);
	*slope_y = ErfInv(2.0f * smax(U2, (float)1e-6f) - 1.0f);
}
inline V3 stretch_sample(const Dist& D, V3 wi, float U1, float U2)           // BeckmannSample microfacet.cc:146-170 / TrowbridgeReitzSample :303-324
{
	V3 ws = normalize(mk(D.ax * wi.x, D.ay * wi.y, wi.z));
	float sx, sy;
	if (D.kind == JP_DIST_BECKMANN) BeckmannSample11(ws.z, U1, U2, &sx, &sy); else tr_sample11(ws.z, U1, U2, &sx, &sy);
	float tmp = cosphi(ws) * sx - sinphi(ws) * sy;
	sy = sinphi(ws) * sx + cosphi(ws) * sy;
	sx = tmp;
	sx = D.ax * sx; sy = D.ay * sy;
	return normalize(mk(-sx, -sy, 1.f));
}
inline float dist_D(const Dist& D, V3 wh)                                    // microfacet.cc:175-192
{
	float tan2Theta = tan2t(wh);
	if (std::isinf(tan2Theta)) return 0.;
	const float cos4Theta = (wh.z * wh.z) * (wh.z * wh.z);
	if (D.kind == JP_DIST_BECKMANN)
		return std::exp(-tan2Theta * (cos2phi(wh) / (D.ax * D.ax) + sin2phi(wh) / (D.ay * D.ay))) / (kPi * D.ax * D.ay * cos4Theta);
	float e = (cos2phi(wh) / (D.ax * D.ax) + sin2phi(wh) / (D.ay * D.ay)) * tan2Theta;
	return 1 / (kPi * D.ax * D.ay * cos4Theta * (1 + e) * (1 + e));
}
inline float dist_Lambda(const Dist& D, V3 w)                                // microfacet.cc:194-214
{
	float absTanTheta = std::abs(tant(w));
	if (std::isinf(absTanTheta)) return 0.;
	float alpha = std::sqrt(cos2phi(w) * D.ax * D.ax + sin2phi(w) * D.ay * D.ay);
	if (D.kind == JP_DIST_BECKMANN)
	{
		float a = 1 / (alpha * absTanTheta);
		if (a >= 1.6f) return 0;
		return (1 - 1.259f * a + 0.396f * a * a) / (3.535f * a + 2.181f * a * a);
	}
	float alpha2Tan2Theta = (alpha * absTanTheta) * (alpha * absTanTheta);
	return (-1 + std::sqrt(1.f + alpha2Tan2Theta)) / 2;
}
inline float dist_G1(const Dist& D, V3 w) { return 1 / (1 + dist_Lambda(D, w)); }                             // microfacet.h:22-25
inline float dist_G(const Dist& D, V3 wo, V3 wi) { return 1 / (1 + dist_Lambda(D, wo) + dist_Lambda(D, wi)); } // microfacet.h:26-28
inline float dist_Pdf(const Dist& D, V3 wo, V3 wh)                           // microfacet.cc:359-365
{
	if (D.vis) return dist_D(D, wh) * dist_G1(D, wo) * absdot(wo, wh) / std::abs(wo.z);
	return dist_D(D, wh) * std::abs(wh.z);
}
inline V3 spherical(float sinTheta, float cosTheta, float phi) { return mk(sinTheta * std::cos(phi), sinTheta * std::sin(phi), cosTheta); }   // geometry.h:203-209
inline V3 dist_sample_wh(const Dist& D, V3 wo, float u0, float u1)           // microfacet.cc:216-254 (Beckmann), :326-357 (TrowbridgeReitz)
{
	if (D.vis)
	{
		bool flip = wo.z < 0;
		V3 wh = stretch_sample(D, flip ? -wo : wo, u0, u1);
		if (flip) wh = -wh;
		return wh;
	}
	V3 wh;
	if (D.kind == JP_DIST_BECKMANN)
	{
		float tan2Theta, phi;
		if (D.ax == D.ay)
		{
			float logSample = std::log(1 - u0);
			tan2Theta = -D.ax * D.ax * logSample;
			phi = u1 * 2 * kPi;
		}
		else
		{
			float logSample = std::log(1 - u0);
			phi = std::atan(D.ay / D.ax * std::tan(2 * kPi * u1 + 0.5f * kPi));
			if (u1 > 0.5f) phi += kPi;
			float sinPhi = std::sin(phi), cosPhi = std::cos(phi);
			float ax2 = D.ax * D.ax, ay2 = D.ay * D.ay;
			tan2Theta = -logSample / (cosPhi * cosPhi / ax2 + sinPhi * sinPhi / ay2);
		}
		float cosTheta = 1 / std::sqrt(1 + tan2Theta);
		float sinTheta = std::sqrt(smax((float)0, 1 - cosTheta * cosTheta));
		wh = spherical(sinTheta, cosTheta, phi);
	}
	else
	{
		float cosTheta = 0, phi = (2 * kPi) * u1;
		if (D.ax == D.ay)
		{
			float tanTheta2 = D.ax * D.ax * u0 / (1.0f - u0);
			cosTheta = 1 / std::sqrt(1 + tanTheta2);
		}
		else
		{
			phi = std::atan(D.ay / D.ax * std::tan(2 * kPi * u1 + .5f * kPi));
			if (u1 > .5f) phi += kPi;
			float sinPhi = std::sin(phi), cosPhi = std::cos(phi);
			const float ax2 = D.ax * D.ax, ay2 = D.ay * D.ay;
			const float alpha2 = 1 / (cosPhi * cosPhi / ax2 + sinPhi * sinPhi / ay2);
			float tanTheta2 = alpha2 * u0 / (1 - u0);
			cosTheta = 1 / std::sqrt(1 + tanTheta2);
		}
		float sinTheta = std::sqrt(smax((float)0., (float)1. - cosTheta * cosTheta));
		wh = spherical(sinTheta, cosTheta, phi);
	}
	if (!same_hemi(wo, wh)) wh = -wh;
	return wh;
}
inline V3 fresnel_of(const JpBsdfDesc& d, float cosI)                        // bsdf.cc:15-24, bsdf.h:664-667
{
	if (d.fresnel == JP_FRESNEL_NOOP) return splat(1.f);
	if (d.fresnel == JP_FRESNEL_DIELECTRIC) return splat(fresnel_dielectric(cosI, d.fr_eta_i[0], d.fr_eta_t[0]));
	return fresnel_conductor(std::abs(cosI), ld3(d.fr_eta_i), ld3(d.fr_eta_t), ld3(d.fr_k));
}

struct Out { V3 f; float pdf; };
// Evalf_Local / Pdf_Local
inline V3 x_eval(const JpBsdfDesc& d, const Frame& fr, V3 wo, V3 wi);
inline float x_pdf(const JpBsdfDesc& d, const Frame& fr, V3 wo, V3 wi)
{
	switch (d.kind)
	{
	case JP_BSDF_LAMBERT: return same_hemi(wo, wi) ? std::abs(wi.z) * kInvPi : 0;                    // bsdf.h:357-360
	case JP_BSDF_MICROFACET_REFLECTION:                                                              // bsdf.cc:53-58
	{
		if (!same_hemi(wo, wi)) return 0;
		Dist D = make_dist(d);
		V3 wh = normalize(wo + wi);
		return dist_Pdf(D, wo, wh) / (4 * dot(wo, wh));
	}
	case JP_BSDF_MICROFACET_TRANSMISSION:                                                            // bsdf.cc:110-124
	{
		if (same_hemi(wo, wi)) return 0;
		Dist D = make_dist(d);
		float eta = wo.z > 0 ? (d.eta_b / d.eta_a) : (d.eta_a / d.eta_b);
		V3 wh = normalize(wo + wi * eta);
		if (dot(wo, wh) * dot(wi, wh) > 0) return 0;
		float sqrtDenom = dot(wo, wh) + eta * dot(wi, wh);
		float dwh_dwi = std::abs((eta * eta * dot(wi, wh)) / (sqrtDenom * sqrtDenom));
		return dist_Pdf(D, wo, wh) * dwh_dwi;
	}
	case JP_BSDF_PHONG:                                                                              // bsdf.h:584-590, 622-626
	{
		const V3 wr = reflect(wo, mk(0, 0, 1));
		const float cosTheta = smax((float)0, dot(wr, wi));
		return (d.exponent + 1) * std::pow(cosTheta, d.exponent) * kInv2Pi;
	}
	default: return 0;                                                                               // delta BSDFs bsdf.h:410-413, 473-476
	}
}
inline V3 x_eval(const JpBsdfDesc& d, const Frame& fr, V3 wo, V3 wi)
{
	switch (d.kind)
	{
	case JP_BSDF_LAMBERT: return same_hemi(wo, wi) ? ld3(d.color) * kInvPi : splat(0);
	case JP_BSDF_MICROFACET_REFLECTION:                                                              // bsdf.cc:35-51
	{
		Dist D = make_dist(d);
		float cosO = std::abs(wo.z), cosI = std::abs(wi.z);
		V3 wh = wi + wo;
		if (cosI == 0 || cosO == 0) return splat(0);
		if (wh.x == 0 && wh.y == 0 && wh.z == 0) return splat(0);
		wh = normalize(wh);
		V3 ff = (dot(wh, mk(0, 0, 1)) < 0) ? -wh : wh;
		V3 F = fresnel_of(d, dot(wi, ff));
		return cmul(ld3(d.color) * dist_D(D, wh) * dist_G(D, wo, wi), F) / (4 * cosI * cosO);
	}
	case JP_BSDF_MICROFACET_TRANSMISSION:                                                            // bsdf.cc:85-108
	{
		if (same_hemi(wo, wi)) return splat(0);
		Dist D = make_dist(d);
		float cosO = wo.z, cosI = wi.z;
		if (cosI == 0 || cosO == 0) return splat(0);
		float eta = wo.z > 0 ? (d.eta_b / d.eta_a) : (d.eta_a / d.eta_b);
		V3 wh = normalize(wo + wi * eta);
		if (wh.z < 0) wh = -wh;
		if (dot(wo, wh) * dot(wi, wh) > 0) return splat(0);
		V3 F = splat(fresnel_dielectric(dot(wo, wh), d.eta_a, d.eta_b));
		float sqrtDenom = dot(wo, wh) + eta * dot(wi, wh);
		float factor = (1 / eta);
		return cmul(splat(1) - F, ld3(d.color)) *
			std::abs(dist_D(D, wh) * dist_G(D, wo, wi) * eta * eta * absdot(wi, wh) * absdot(wo, wh) * factor * factor / (cosI * cosO * sqrtDenom * sqrtDenom));
	}
	case JP_BSDF_PHONG:                                                                              // bsdf.h:571-582
	{
		if (!same_hemi(wo, wi)) return splat(0);
		const V3 wr = reflect(wo, mk(0, 0, 1));
		const float cos_alpha = dot(wr, wi);
		const V3 rho = ld3(d.color) * (d.exponent + 2.f) * kInv2Pi;
		return rho * std::pow(cos_alpha, d.exponent);
	}
	default: return splat(0);
	}
}
inline BsdfSample x_sample(const JpBsdfDesc& d, const Frame& fr, V3 wo, float ux, float uy)
{
	BsdfSample s = empty_sample();
	switch (d.kind)
	{
	case JP_BSDF_LAMBERT: case JP_BSDF_MIRROR: case JP_BSDF_FRESNEL_SPECULAR:
	{   // the closures the materials build: the path's own code
		Closure c; c.frame = fr; c.c0 = ld3(d.color); c.c1 = ld3(d.color2); c.eta_i = d.eta_a; c.eta_t = d.eta_b; c.ax = c.ay = 0; c.fresnel = FR_CONDUCTOR; c.feta = c.fk = splat(0);
		c.kind = d.kind == JP_BSDF_LAMBERT ? CL_LAMBERT : (d.kind == JP_BSDF_MIRROR ? CL_MIRROR : CL_FRESNEL_SPECULAR);
		return sample_local(c, wo, ux, uy);
	}
	case JP_BSDF_MICROFACET_REFLECTION:                                                              // bsdf.cc:60-78
	{
		if (wo.z == 0) return s;
		Dist D = make_dist(d);
		V3 wh = dist_sample_wh(D, wo, ux, uy);
		if (dot(wo, wh) < 0) return s;
		V3 wi = reflect(wo, wh);
		if (!same_hemi(wo, wi)) return s;
		s.wi = wi;
		s.f = x_eval(d, fr, wo, wi);
		s.pdf = dist_Pdf(D, wo, wh) / (4 * dot(wo, wh));
		s.flags = BS_REFLECTION | BS_GLOSSY;
		return s;
	}
	case JP_BSDF_MICROFACET_TRANSMISSION:                                                            // bsdf.cc:126-145
	{
		if (wo.z == 0) return s;
		Dist D = make_dist(d);
		V3 wh = dist_sample_wh(D, wo, ux, uy);
		if (dot(wo, wh) < 0) return s;
		V3 wi;
		float eta = wo.z > 0 ? (d.eta_a / d.eta_b) : (d.eta_b / d.eta_a);
		if (!refract(wo, wh, eta, &wi)) return s;
		s.wi = wi;
		// bsdf.cc:141 calls FBSDF::Pdf -- the WORLD-space entry (bsdf.h:290-293) -- on the local vectors: they go through ToLocal once more
		s.pdf = x_pdf(d, fr, to_local(fr, wo), to_local(fr, wi));
		s.f = x_eval(d, fr, wo, wi);
		s.flags = BS_TRANSMISSION | BS_GLOSSY;
		return s;
	}
	case JP_BSDF_PHONG:                                                                              // bsdf.h:592-611
	{
		const float phi = 2 * kPi * ux;
		const float cos_theta = std::pow(uy, (float)1 / (d.exponent + 1));
		const float sin_theta = std::sqrt(1.f - cos_theta * cos_theta);
		V3 wl = mk(std::cos(phi) * sin_theta, std::sin(phi) * sin_theta, cos_theta);
		const V3 wr = reflect(wo, mk(0, 0, 1));
		Frame lobe = frame_from_z(wr);
		s.wi = to_world(lobe, wl);
		if (wo.z < 0) s.wi.z *= -1;
		s.f = x_eval(d, fr, wo, s.wi);
		s.pdf = x_pdf(d, fr, wo, s.wi);
		s.flags = BS_REFLECTION | BS_GLOSSY;
		return s;
	}
	}
	return s;
}
} // namespace xb

// FMaterial::Scattering (material.h:34-37, 52-55, 72-75; material.cc:12-43).  Returns false for a null material.
inline bool scattering(const JpScene* js, int mat, V3 normal, Draw& rnd, Closure& c)
{
	if (mat < 0) return false;
	const float* p = js->mat_params + (size_t)mat * JP_MAT_PARAM_STRIDE;
	int type = js->mat_type[mat];
	c.c0 = c.c1 = splat(0); c.eta_i = c.eta_t = 1; c.ax = c.ay = 0; c.fresnel = FR_CONDUCTOR; c.feta = c.fk = splat(0);
	if (type == JP_MAT_PLASTIC)
	{
		float u = rnd.get1();                                              // material.cc:14 (drawn before the frame is built)
		float Qd = p[7];
		c.frame = frame_from_z(normal);
		if (u < Qd) { c.kind = CL_LAMBERT; c.c0 = ld3(p) / Qd; }
		else { c.kind = CL_MICROFACET; c.c0 = ld3(p + 3) / (1 - Qd); c.fresnel = FR_DIELECTRIC; c.ax = c.ay = smax(0.001f, p[6]); }   // microfacet.h:72-74
		return true;
	}
	c.frame = frame_from_z(normal);
	switch (type)
	{
	case JP_MAT_MATTE: c.kind = CL_LAMBERT; c.c0 = ld3(p); break;
	case JP_MAT_MIRROR: c.kind = CL_MIRROR; c.c0 = ld3(p); break;
	case JP_MAT_GLASS: c.kind = CL_FRESNEL_SPECULAR; c.eta_i = 1.f; c.eta_t = p[0]; c.c0 = ld3(p + 1); c.c1 = ld3(p + 4); break;
	case JP_MAT_METAL: c.kind = CL_MICROFACET; c.c0 = splat(1.f); c.fresnel = FR_CONDUCTOR; c.feta = ld3(p); c.fk = ld3(p + 3);
		c.ax = smax(0.001f, p[6]); c.ay = smax(0.001f, p[7]); break;
	default: return false;
	}
	return true;
}

// ---------------------------------------------------------------------------------------------
// lights (light.h, shape.h sampling)
// ---------------------------------------------------------------------------------------------
struct LightSample { V3 pos, wi; float pdf; V3 Li; };

inline float shape_area(const Scene& sc, int prim)
{
	int t = sc.js->prim_shape_type[prim], i = sc.js->prim_shape_index[prim];
	if (t == JP_SHAPE_TRIANGLE) return 0.5f * len(cross(sc.tris[i].p1 - sc.tris[i].p0, sc.tris[i].p2 - sc.tris[i].p0));     // shape.h:351
	if (t == JP_SHAPE_RECTANGLE) return len(cross(sc.rects[i].p0 - sc.rects[i].p1, sc.rects[i].p2 - sc.rects[i].p1));       // shape.h:457
	if (t == JP_SHAPE_DISK) return kPi * sc.disks[i].r * sc.disks[i].r;                                                     // shape.h:253
	return 4 * kPi * (sc.sphs[i].r * sc.sphs[i].r);                                                                         // shape.h:546 (radius2 = r*r)
}

// FShape::SamplePosition overrides: shape.h:353-363 (triangle), :459-467 (rectangle), :549-562 (sphere)
inline void sample_position(const Scene& sc, int prim, float ux, float uy, V3& pos, V3& nrm, float& pdf)
{
	int t = sc.js->prim_shape_type[prim], i = sc.js->prim_shape_index[prim];
	if (t == JP_SHAPE_TRIANGLE)
	{
		const Tri& T = sc.tris[i];
		float su0 = std::sqrt(ux); float bx = 1 - su0, by = uy * su0;         // sampling.h:121-125
		pos = bx * T.p0 + by * T.p1 + (1 - bx - by) * T.p2; nrm = T.n;
	}
	else if (t == JP_SHAPE_RECTANGLE)
	{
		const Rect& R = sc.rects[i];
		pos = R.p1 + (R.p0 - R.p1) * ux + (R.p2 - R.p1) * uy; nrm = R.n;
	}
	else if (t == JP_SHAPE_DISK)                                              // shape.h:256-268
	{
		const Disk& D = sc.disks[i];
		Frame fr = frame_from_z(D.n);
		float px, py; concentric_disk(ux, uy, px, py);
		pos = D.c + D.r * (fr.s * px + fr.t * py); nrm = D.n;
	}
	else
	{
		const Sph& S = sc.sphs[i];
		V3 dir = uniform_sphere(ux, uy);
		pos = S.c + S.r * dir; nrm = normalize(dir);
	}
	pdf = 1 / shape_area(sc, prim);
}

// FShape::SampleDirection default shape.h:124-145 and the FSphere override shape.h:564-644
inline void sample_direction(const Scene& sc, int prim, const Isect& is, float ux, float uy, V3& pos, V3& nrm, float& pdf)
{
	int t = sc.js->prim_shape_type[prim], i = sc.js->prim_shape_index[prim];
	if (t != JP_SHAPE_SPHERE)
	{
		sample_position(sc, prim, ux, uy, pos, nrm, pdf);
		V3 wi = pos - is.p;
		float dist2 = len2(wi);
		if (dist2 == 0) pdf = 0;
		else
		{
			wi = normalize(wi);
			pdf *= dist2 / absdot(nrm, -wi);
			if (std::isinf(pdf)) pdf = 0;
		}
		return;
	}
	const Sph& S = sc.sphs[i];
	if (len2(is.p - S.c) <= S.r * S.r)                                        // shape.h:567-586 (inside / on the sphere)
	{
		sample_position(sc, prim, ux, uy, pos, nrm, pdf);
		V3 wi = pos - is.p;
		if (len2(wi) == 0) pdf = 0;
		else
		{
			wi = normalize(wi);
			pdf *= len2(pos - is.p) / absdot(is.n, -wi);                      // uses isect.normal (shape.h:579)
		}
		if (std::isinf(pdf)) pdf = 0;
		return;
	}
	float dist = len(is.p - S.c);                                             // shape.h:603-643
	float inv_dist = 1 / dist;
	float sin_max = S.r * inv_dist;
	float sin_max2 = sin_max * sin_max;
	float inv_sin_max = 1 / sin_max;
	float cos_max = std::sqrt(smax(0.f, 1 - sin_max2));
	float cos_theta = (cos_max - 1) * ux + 1;
	float sin_theta2 = 1 - cos_theta * cos_theta;
	if (sin_max2 < 0.00068523f)
	{
		sin_theta2 = sin_max2 * ux;
		cos_theta = std::sqrt(1 - sin_theta2);
	}
	float cos_alpha = sin_theta2 * inv_sin_max + cos_theta * std::sqrt(smax(0.f, 1.f - sin_theta2 * inv_sin_max * inv_sin_max));
	float sin_alpha = std::sqrt(smax(0.f, 1.f - cos_alpha * cos_alpha));
	float phi = uy * 2 * kPi;
	V3 normal = (S.c - is.p) * inv_dist;
	Frame fr = frame_from_z(normal);
	// Spherical_2_Direction(sin, cos, phi, x, y, z) geometry.h:201-210 with (-s, -t, -n)
	V3 wn = sin_alpha * std::cos(phi) * (-fr.s) + sin_alpha * std::sin(phi) * (-fr.t) + cos_alpha * (-fr.n);
	pos = S.c + S.r * mk(wn.x, wn.y, wn.z);
	nrm = wn;
	pdf = 1 / (2 * kPi * (1 - cos_max));
}

inline LightSample sample_li(const Scene& sc, int li, const Isect& is, float ux, float uy)
{
	const JpScene* js = sc.js;
	LightSample s; s.pos = mk(0, 0, 0); s.wi = mk(0, 0, 0); s.pdf = 0; s.Li = splat(0);
	V3 radiance = ld3(js->light_radiance + 3 * li);
	if (js->light_type[li] == JP_LIGHT_ENVIRONMENT)                           // light.h:265-291
	{
		float theta = uy * kPi, phi = ux * 2 * kPi;
		float cosT = std::cos(theta), sinT = std::sin(theta);
		float sinP = std::sin(phi), cosP = std::cos(phi);
		s.wi = mk(sinT * cosP, sinT * sinP, cosT);
		s.pos = is.p + s.wi * 2 * js->world_radius;
		s.pdf = 0;
		if (sinT != 0) s.pdf = 1 / (2 * kPi * kPi * sinT);
		s.Li = radiance;
		return s;
	}
	if (js->light_type[li] == JP_LIGHT_POINT)                                 // FPointLight::Sample_Li light.h:94-123
	{
		V3 wp = ld3(js->light_vec + 3 * li);
		s.pos = wp;
		s.wi = normalize(wp - is.p);
		s.pdf = 1.f;
		s.Li = radiance / len2(wp - is.p);
		return s;
	}
	if (js->light_type[li] == JP_LIGHT_DIRECTION)                             // FDirectionLight::Sample_Li light.h:155-164
	{
		s.wi = -ld3(js->light_vec + 3 * li);
		s.pos = is.p + s.wi * 2 * js->world_radius;
		s.pdf = 1.f;
		s.Li = radiance;
		return s;
	}
	V3 lp, ln;                                                                // FAreaLight::Sample_Li light.h:199-216
	sample_direction(sc, js->light_prim[li], is, ux, uy, lp, ln, s.pdf);
	s.pos = lp;
	if (s.pdf == 0 || len2(lp - is.p) == 0) s.Li = splat(0);
	else
	{
		s.wi = normalize(lp - is.p);
		s.Li = (dot(ln, -s.wi) > 0.f) ? radiance : splat(0);                  // FAreaLight::L light.h:234-238
	}
	return s;
}

// ---------------------------------------------------------------------------------------------
// integrator (integrator.cc:82-111, 316-403)
// ---------------------------------------------------------------------------------------------
struct Counts { unsigned long long closest, closest_hit, shadow, shadow_occ; };

inline bool occluded(const Scene& sc, const Isect& is, V3 target, Counts& cn)   // scene.h:36-47
{
	V3 dir = normalize(target - is.p);
	float dist = len(is.p - target);
	Ray r = mkray(is.p, dir, 0.001f, dist - 0.001f);
	Isect unused; unused.prim = -1;
	bool hit = sc.intersect(r, unused);
	cn.shadow++; if (hit) cn.shadow_occ++;
	return hit;
}

V3 Li(const Scene& sc, const Ray& inRay, Draw& rnd, int maxDepth, Counts& cn)   // integrator.cc:316-403
{
	const JpScene* js = sc.js;
	V3 L = mk(0, 0, 0), beta = mk(1, 1, 1);
	Ray ray = inRay;
	bool specular = false;
	for (int bounces = 0;; ++bounces)
	{
		Isect is; is.prim = -1;
		bool found = sc.intersect(ray, is);
		cn.closest++; if (found) cn.closest_hit++;
		if (bounces == 0 || specular)
		{
			if (found)
			{
				int li = js->prim_light[is.prim];                                // primitive.h:60-63
				V3 Le = (li >= 0 && dot(is.n, is.wo) > 0.f) ? ld3(js->light_radiance + 3 * li) : splat(0);
				L = L + cmul(beta, Le);
			}
			else for (size_t k = 0; k < sc.infiniteLights.size(); k++) L = L + cmul(beta, ld3(js->light_radiance + 3 * sc.infiniteLights[k]));   // light.h:300-303
		}
		if (!found || bounces >= maxDepth) break;
		const V3 N = is.n;
		Closure c;
		if (!scattering(js, js->prim_material[is.prim], is.n, rnd, c))            // integrator.cc:348-353
		{
			ray = mkray(is.p, ray.d);
			--bounces;
			continue;
		}
		if (!c.delta())
		{
			for (int li = 0; li < js->n_lights; li++)                             // integrator.cc:359-371
			{
				float ux, uy; rnd.get2(ux, uy);
				LightSample ls = sample_li(sc, li, is, ux, uy);
				if (isblack(ls.Li) || ls.pdf == 0.f) continue;
				V3 f = bsdf_eval(c, is.wo, ls.wi);
				if (!isblack(f) && !occluded(sc, is, ls.pos, cn))
					L = L + cmul(cmul(beta, f), ls.Li) * absdot(ls.wi, N) / ls.pdf;
			}
		}
		float ux, uy; rnd.get2(ux, uy);
		BsdfSample bs = bsdf_sample(c, is.wo, ux, uy);                            // integrator.cc:375-379
		if (isblack(bs.f) || bs.pdf == 0.f) break;
		specular = (bs.flags & BS_SPECULAR) != 0;
		if (bounces >= 3)                                                         // integrator.cc:383-393
		{
			float q = smax(0.05f, 1 - maxcomp(bs.f));
			if (rnd.get1() < q) break;
			beta = cmul(beta, bs.f * absdot(bs.wi, is.n) / (bs.pdf * (1 - q)));
			ray = mkray(is.p, bs.wi);
		}
		else
		{
			beta = cmul(beta, bs.f * absdot(bs.wi, is.n) / bs.pdf);
			ray = mkray(is.p, bs.wi);
		}
	}
	return L;
}

// FWhittedIntegrator::Li integrator.cc:115-220.  typeFlags: bsdf.h:340, 398, 460, 561; MatchTypes bsdf.h:282
inline int closure_type_flags(const Closure& c)
{
	return c.kind == CL_LAMBERT ? (1 | 8) : c.kind == CL_MIRROR ? (1 | 4) : c.kind == CL_FRESNEL_SPECULAR ? (4 | 1 | 2) : (1 | 16);
}
V3 LiWhitted(const Scene& sc, const Ray& ray, Draw& rnd, int maxDepth, int depth, Counts& cn)
{
	const JpScene* js = sc.js;
	V3 L = mk(0, 0, 0);
	Isect is; is.prim = -1;
	bool found = sc.intersect(ray, is);
	cn.closest++; if (found) cn.closest_hit++;
	if (!found)
	{
		for (size_t k = 0; k < sc.infiniteLights.size(); k++) L = L + ld3(js->light_radiance + 3 * sc.infiniteLights[k]);
		return L;
	}
	const V3 N = is.n;
	Closure c;
	if (!scattering(js, js->prim_material[is.prim], is.n, rnd, c))
		return LiWhitted(sc, mkray(is.p, ray.d), rnd, maxDepth, depth, cn);           // integrator.cc:137-139
	{
		int li = js->prim_light[is.prim];
		V3 Le = (li >= 0 && dot(is.n, is.wo) > 0.f) ? ld3(js->light_radiance + 3 * li) : splat(0);
		L = L + Le;                                                                   // integrator.cc:142
	}
	for (int li = 0; li < js->n_lights; li++)                                         // integrator.cc:145-158
	{
		float ux, uy; rnd.get2(ux, uy);
		LightSample ls = sample_li(sc, li, is, ux, uy);
		if (isblack(ls.Li) || ls.pdf == 0.f) continue;
		V3 f = bsdf_eval(c, is.wo, ls.wi);
		if (!isblack(f) && !occluded(sc, is, ls.pos, cn))
			L = L + cmul(f, ls.Li) * absdot(ls.wi, N) / ls.pdf;
	}
	if (depth + 1 < maxDepth)                                                         // integrator.cc:161-167
	{
		const int flags = closure_type_flags(c);
		const int want[3] = { 4 | 1, 4 | 2, 4 | 1 | 2 };                              // SpecularReflect, SpecularTransmit, SpecularReflectAndTransmit
		for (int k = 0; k < 3; k++)
		{
			V3 add = splat(0);
			if ((flags & want[k]) == flags)
			{
				float ux, uy; rnd.get2(ux, uy);
				BsdfSample bs = bsdf_sample(c, is.wo, ux, uy);
				if (!(isblack(bs.f) || bs.pdf == 0.f))
					add = cmul(bs.f, LiWhitted(sc, mkray(is.p, bs.wi), rnd, maxDepth, depth + 1, cn)) * absdot(bs.wi, is.n) / bs.pdf;
			}
			L = L + add;
		}
	}
	return L;
}

// FDebugIntegrator::Li integrator.h:44-58: the hit normal as a colour
V3 LiDebug(const Scene& sc, const Ray& ray, Counts& cn)
{
	Isect is; is.prim = -1;
	bool found = sc.intersect(ray, is);
	cn.closest++; if (found) cn.closest_hit++;
	return found ? is.n : mk(0, 0, 0);
}

inline Ray camera_ray(const JpCamera& c, float fx, float fy)                      // camera.h:52-58
{
	V3 dir = ld3(c.front) + ld3(c.right) * (fx / c.res_x - 0.5f) + ld3(c.up) * (0.5f - fy / c.res_y);
	return mkray(ld3(c.pos), normalize(dir));
}

// FIntegrator::DoRender integrator.cc:82-111 over rows [y0,y1)
void render_rows(const Scene& sc, const JpRenderParams& rp, Sampler& smp, int y0, int y1, float* film, Counts& cn)
{
	float ratio = (float)1 / rp.spp;
	Draw rnd = { &smp, nullptr };
	for (int y = y0; y < y1; y++) for (int x = 0; x < rp.width; x++)
	{
		V3 L = mk(0, 0, 0);
		smp.start_pixel();
		do
		{
			float fx, fy; smp.camera_sample((float)x, (float)y, fx, fy);
			Ray ray = camera_ray(sc.js->camera, fx, fy);
			const V3 Ls = rp.integrator == JP_INTEGRATOR_WHITTED ? LiWhitted(sc, ray, rnd, rp.max_depth, 0, cn)
			            : rp.integrator == JP_INTEGRATOR_DEBUG_NORMAL ? LiDebug(sc, ray, cn) : Li(sc, ray, rnd, rp.max_depth, cn);
			V3 dL = Ls * ratio;
			L = L + dL;
		} while (smp.next_sample());
		float* o = film + 3 * ((size_t)y * rp.width + x);                         // film.h:22-23, 64-68 onto a zero film
		o[0] = 0.f + clampf(L.x, 0.f, 1.f); o[1] = 0.f + clampf(L.y, 0.f, 1.f); o[2] = 0.f + clampf(L.z, 0.f, 1.f);
	}
}

} // namespace

extern "C" {

// FIntegrator::Render integrator.cc:35-80: nthreads < 1 -> serial whole-frame path with ONE sampler stream;
// otherwise 20-row bands (rp->band_rows), a fresh sampler per band, worker threads pulling bands.
int jp_oracle_render(const JpScene* js, const JpRenderParams* rp, int nthreads, float* film, JpCounters* out)
{
	if (!js || !rp || !film) return JP_ERR_INVALID_ARGUMENT;
	Scene sc(js);
	const int W = rp->width, H = rp->height;
	std::memset(film, 0, sizeof(float) * 3 * (size_t)W * H);
	Counts total = { 0, 0, 0, 0 };
	if (nthreads < 1)
	{
		Sampler smp(rp->sampler_mode, rp->seed, rp->spp);
		render_rows(sc, *rp, smp, 0, H, film, total);
	}
	else
	{
		const int rows = rp->band_rows > 0 ? rp->band_rows : 20;
		const int nbands = (H + rows - 1) / rows;
		std::atomic<int> next(0);
		std::vector<Counts> per(nthreads, total);
		std::vector<std::thread> pool;
		for (int t = 0; t < nthreads; t++) pool.push_back(std::thread([&, t]() {
			for (;;)
			{
				int b = next.fetch_add(1);
				if (b >= nbands) break;
				if (rp->shard_count > 1 && (b % rp->shard_count) != rp->shard_index) continue;
				Sampler smp(rp->sampler_mode, rp->seed, rp->spp);                   // sampler->Clone() integrator.cc:66
				int y0 = b * rows, y1 = y0 + rows; if (y1 > H) y1 = H;
				render_rows(sc, *rp, smp, y0, y1, film, per[t]);
			}
		}));
		for (size_t t = 0; t < pool.size(); t++) pool[t].join();
		for (int t = 0; t < nthreads; t++) { total.closest += per[t].closest; total.closest_hit += per[t].closest_hit; total.shadow += per[t].shadow; total.shadow_occ += per[t].shadow_occ; }
	}
	if (out)
	{
		std::memset(out, 0, sizeof(*out));
		out->closest_rays = total.closest; out->closest_hits = total.closest_hit; out->shadow_rays = total.shadow; out->shadow_occluded = total.shadow_occ;
	}
	return JP_OK;
}

void* jp_oracle_scene_new(const JpScene* js) { return new Scene(js); }
void  jp_oracle_scene_free(void* h) { delete (Scene*)h; }
void  jp_oracle_set_watertight(int on) { g_watertight = on != 0; }
// debug: the tree in preorder, an interior node whose only child is a leaf reported as that leaf.  boxes: 6 floats per node,
// kind: -1 interior, else the number of objects of the leaf; order: the objects of the leaves in visiting order.
static void dbg_dump(const Scene& sc, int n, std::vector<float>& boxes, std::vector<int>& kind, std::vector<int>& order)
{
	const Node& nd = sc.nodes[n];
	const bool single = !nd.leaf && nd.right < 0;
	const Node& use = single ? sc.nodes[nd.left] : nd;
	boxes.push_back(nd.box.mn.x); boxes.push_back(nd.box.mn.y); boxes.push_back(nd.box.mn.z); boxes.push_back(nd.box.mx.x); boxes.push_back(nd.box.mx.y); boxes.push_back(nd.box.mx.z);
	if (use.leaf) { kind.push_back(use.count); for (int i = 0; i < use.count; i++) order.push_back(sc.order[use.first + i]); return; }
	kind.push_back(-1);
	dbg_dump(sc, nd.left, boxes, kind, order);
	if (nd.right >= 0) dbg_dump(sc, nd.right, boxes, kind, order);
}
int jp_oracle_tree_dump(void* h, float* boxes, int* kind, int* order, int max_nodes)
{
	const Scene& sc = *(Scene*)h;
	std::vector<float> b; std::vector<int> k, o;
	if (sc.root >= 0) dbg_dump(sc, sc.root, b, k, o);
	if ((int)k.size() > max_nodes) return -(int)k.size();
	std::memcpy(boxes, b.data(), b.size() * sizeof(float)); std::memcpy(kind, k.data(), k.size() * sizeof(int)); std::memcpy(order, o.data(), o.size() * sizeof(int));
	return (int)k.size();
}

// debug: the chain of nodes from the root to the leaf that holds `prim`; for each, its box and whether the box test lets
// the ray in.  out: 8 floats per level (min xyz, max xyz, box test result, is leaf); returns the number of levels.
static bool dbg_find(const Scene& sc, int n, int prim, std::vector<int>& path)
{
	const Node& nd = sc.nodes[n];
	path.push_back(n);
	if (nd.leaf) { for (int i = 0; i < nd.count; i++) if (sc.order[nd.first + i] == prim) return true; path.pop_back(); return false; }
	if (dbg_find(sc, nd.left, prim, path)) return true;
	if (nd.right >= 0 && dbg_find(sc, nd.right, prim, path)) return true;
	path.pop_back(); return false;
}
int jp_oracle_debug_path(void* h, const float* o, const float* d, float tmin, float tmax, int prim, float* out, int max_levels)
{
	const Scene& sc = *(Scene*)h;
	std::vector<int> path; if (!dbg_find(sc, sc.root, prim, path)) return -1;
	int k = 0;
	for (int n : path)
	{
		if (k >= max_levels) break;
		const Node& nd = sc.nodes[n];
		Ray r = mkray(ld3(o), ld3(d), tmin, tmax);
		out[8 * k + 0] = nd.box.mn.x; out[8 * k + 1] = nd.box.mn.y; out[8 * k + 2] = nd.box.mn.z;
		out[8 * k + 3] = nd.box.mx.x; out[8 * k + 4] = nd.box.mx.y; out[8 * k + 5] = nd.box.mx.z;
		out[8 * k + 6] = nd.leaf ? 1.f : (box_hit(nd.box, r) ? 1.f : 0.f); out[8 * k + 7] = nd.leaf ? 1.f : 0.f;
		k++;
	}
	return k;
}

void jp_oracle_trace(void* h, int n, const float* o, const float* d, const float* tmin, const float* tmax,
                     int* hit, float* t, int* prim, float* nrm, float* pos)
{
	const Scene& sc = *(Scene*)h;
	for (int i = 0; i < n; i++)
	{
		Ray r = mkray(ld3(o + 3 * i), ld3(d + 3 * i), tmin[i], tmax[i]);
		Isect is; is.prim = -1; is.n = mk(0, 0, 0); is.p = mk(0, 0, 0);
		bool b = sc.intersect(r, is);
		hit[i] = b; t[i] = r.tmax; prim[i] = b ? is.prim : -1;
		nrm[3 * i] = b ? is.n.x : 0; nrm[3 * i + 1] = b ? is.n.y : 0; nrm[3 * i + 2] = b ? is.n.z : 0;
		pos[3 * i] = b ? is.p.x : 0; pos[3 * i + 1] = b ? is.p.y : 0; pos[3 * i + 2] = b ? is.p.z : 0;
	}
}

void jp_oracle_camera_rays(void* h, int n, const float* pxy, float* o, float* d)
{
	const Scene& sc = *(Scene*)h;
	for (int i = 0; i < n; i++)
	{
		Ray r = camera_ray(sc.js->camera, pxy[2 * i], pxy[2 * i + 1]);
		o[3 * i] = r.o.x; o[3 * i + 1] = r.o.y; o[3 * i + 2] = r.o.z; d[3 * i] = r.d.x; d[3 * i + 1] = r.d.y; d[3 * i + 2] = r.d.z;
	}
}

void jp_oracle_bsdf(void* h, int count, int mat, const float* n, const float* wo, const float* wi, const float* u2, const float* uscat,
                    float* feval, float* sf, float* swi, float* spdf, int* sflags, int* isdelta)
{
	const Scene& sc = *(Scene*)h;
	for (int i = 0; i < count; i++)
	{
		Script k = { uscat + i, 1, 0 }; Draw rnd = { nullptr, &k };
		Closure c; scattering(sc.js, mat, ld3(n + 3 * i), rnd, c);
		V3 f = bsdf_eval(c, ld3(wo + 3 * i), ld3(wi + 3 * i));
		BsdfSample s = bsdf_sample(c, ld3(wo + 3 * i), u2[2 * i], u2[2 * i + 1]);
		feval[3 * i] = f.x; feval[3 * i + 1] = f.y; feval[3 * i + 2] = f.z;
		sf[3 * i] = s.f.x; sf[3 * i + 1] = s.f.y; sf[3 * i + 2] = s.f.z;
		swi[3 * i] = s.wi.x; swi[3 * i + 1] = s.wi.y; swi[3 * i + 2] = s.wi.z;
		spdf[i] = s.pdf; sflags[i] = s.flags; isdelta[i] = c.delta() ? 1 : 0;
	}
}

// FBSDF::Evalf / Pdf / Sample (bsdf.h:284-302) of a BSDF given by value (JpBsdfDesc), frame = FFrame(normal)
void jp_oracle_bsdf_direct(const JpBsdfDesc* d, int count, const float* n, const float* wo, const float* wi, const float* u2,
                           float* feval, float* pdfeval, float* sf, float* swi, float* spdf, int* sflags)
{
	for (int i = 0; i < count; i++)
	{
		const Frame fr = frame_from_z(ld3(n + 3 * i));
		const V3 wol = to_local(fr, ld3(wo + 3 * i)), wil = to_local(fr, ld3(wi + 3 * i));
		V3 f = xb::x_eval(*d, fr, wol, wil);
		float pe = xb::x_pdf(*d, fr, wol, wil);
		BsdfSample s = xb::x_sample(*d, fr, wol, u2[2 * i], u2[2 * i + 1]);
		s.wi = to_world(fr, s.wi);
		feval[3 * i] = f.x; feval[3 * i + 1] = f.y; feval[3 * i + 2] = f.z; pdfeval[i] = pe;
		sf[3 * i] = s.f.x; sf[3 * i + 1] = s.f.y; sf[3 * i + 2] = s.f.z;
		swi[3 * i] = s.wi.x; swi[3 * i + 1] = s.wi.y; swi[3 * i + 2] = s.wi.z;
		spdf[i] = s.pdf; sflags[i] = s.flags;
	}
}

void jp_oracle_light_sample(void* h, int count, int li, const float* p, const float* n, const float* u2,
                            float* pos, float* wi, float* pdf, float* Lo)
{
	const Scene& sc = *(Scene*)h;
	for (int i = 0; i < count; i++)
	{
		Isect is; is.p = ld3(p + 3 * i); is.n = ld3(n + 3 * i); is.wo = mk(0, 0, 1); is.prim = -1;
		LightSample s = sample_li(sc, li, is, u2[2 * i], u2[2 * i + 1]);
		pos[3 * i] = s.pos.x; pos[3 * i + 1] = s.pos.y; pos[3 * i + 2] = s.pos.z;
		wi[3 * i] = s.wi.x; wi[3 * i + 1] = s.wi.y; wi[3 * i + 2] = s.wi.z;
		pdf[i] = s.pdf; Lo[3 * i] = s.Li.x; Lo[3 * i + 1] = s.Li.y; Lo[3 * i + 2] = s.Li.z;
	}
}

void jp_oracle_li_scripted(void* h, int count, int maxdepth, const float* pxy, const float* vals, int nvals, float* out)
{
	const Scene& sc = *(Scene*)h;
	Counts cn = { 0, 0, 0, 0 };
	for (int i = 0; i < count; i++)
	{
		Script k = { vals + (size_t)i * nvals, nvals, 0 }; Draw rnd = { nullptr, &k };
		float ux = k.next(), uy = k.next();
		Ray r = camera_ray(sc.js->camera, pxy[2 * i] + ux, pxy[2 * i + 1] + uy);
		V3 c = Li(sc, r, rnd, maxdepth, cn);
		out[3 * i] = c.x; out[3 * i + 1] = c.y; out[3 * i + 2] = c.z;
	}
}

void jp_oracle_stock_stream(int n, float* out) { Sampler s(JP_SAMPLER_STOCK_MT19937, 0, 1); for (int i = 0; i < n; i++) out[i] = s.stock(); }

} // extern "C"
