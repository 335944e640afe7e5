"""Synthetic scene scripts.  The reference's scene assets are not in its repository
(main.cc:34-53,94-106 read scene\\cornellbox\\*.obj and scene\\bunny\\bunny.obj), so the Cornell-box-shaped
and bunny-shaped scenes are synthesised here: geometry is written as triangulated single-mesh OBJ text
(%.9g, exact float round trip) and the scene is then created through the SAME call sequence as
main.cc:13-111 on any *backend* exposing the procedural scene API:

    camera / envlight / mat_matte / mat_mirror / mat_glass / mat_plastic / mat_metal /
    mesh / rect / sphere / preprocess

HostBackend (below) drives the product's host library; the test suite has an equivalent backend over the
compiled reference, so both sides always receive identical inputs.
"""
import ctypes as C
import hashlib
import os
import tempfile

import numpy as np

f32 = np.float32


def _fa(v):
    a = np.asarray(v, dtype=np.float32)
    return a.ctypes.data_as(C.POINTER(C.c_float)), a


def asset_dir():
    d = os.environ.get("JETPBRT_ASSET_DIR") or os.path.join(tempfile.gettempdir(), "jetpbrt_assets_%d" % os.getuid())
    os.makedirs(d, exist_ok=True)
    return d


def write_obj(path, verts, faces):
    """verts (n,3) float32, faces (m,3) int 0-based -> triangulated single-mesh OBJ."""
    verts = np.asarray(verts, np.float32)
    lines = ["o mesh"]
    lines += ["v %.9g %.9g %.9g" % (float(v[0]), float(v[1]), float(v[2])) for v in verts]
    lines += ["f %d %d %d" % (f[0] + 1, f[1] + 1, f[2] + 1) for f in np.asarray(faces)]
    txt = "\n".join(lines) + "\n"
    tmp = path + ".tmp%d" % os.getpid()
    with open(tmp, "w") as fh:
        fh.write(txt)
    os.replace(tmp, path)
    return path


def quads_to_obj(path, quads):
    """each quad (v0,v1,v2,v3) -> triangles (v0,v1,v2),(v0,v2,v3)."""
    verts, faces = [], []
    for q in quads:
        b = len(verts)
        verts += [q[0], q[1], q[2], q[3]]
        faces += [(b, b + 1, b + 2), (b, b + 2, b + 3)]
    return write_obj(path, np.array(verts, np.float32), np.array(faces))


# ---- canonical Cornell box geometry ---------------------------------------------------
CORNELL = {
    "light": [[(343, 548.7, 227), (343, 548.7, 332), (213, 548.7, 332), (213, 548.7, 227)]],
    "floor": [[(552.8, 0, 0), (0, 0, 0), (0, 0, 559.2), (549.6, 0, 559.2)],
              [(556, 548.8, 0), (556, 548.8, 559.2), (0, 548.8, 559.2), (0, 548.8, 0)],
              [(549.6, 0, 559.2), (0, 0, 559.2), (0, 548.8, 559.2), (556, 548.8, 559.2)]],
    "right": [[(0, 0, 559.2), (0, 0, 0), (0, 548.8, 0), (0, 548.8, 559.2)]],
    "left": [[(552.8, 0, 0), (549.6, 0, 559.2), (556, 548.8, 559.2), (556, 548.8, 0)]],
    "shortbox": [[(130, 165, 65), (82, 165, 225), (240, 165, 272), (290, 165, 114)],
                 [(290, 0, 114), (290, 165, 114), (240, 165, 272), (240, 0, 272)],
                 [(130, 0, 65), (130, 165, 65), (290, 165, 114), (290, 0, 114)],
                 [(82, 0, 225), (82, 165, 225), (130, 165, 65), (130, 0, 65)],
                 [(240, 0, 272), (240, 165, 272), (82, 165, 225), (82, 0, 225)]],
    "tallbox": [[(423, 330, 247), (265, 330, 296), (314, 330, 456), (472, 330, 406)],
                [(423, 0, 247), (423, 330, 247), (472, 330, 406), (472, 0, 406)],
                [(472, 0, 406), (472, 330, 406), (314, 330, 456), (314, 0, 456)],
                [(314, 0, 456), (314, 330, 456), (265, 330, 296), (265, 0, 296)],
                [(265, 0, 296), (265, 330, 296), (423, 330, 247), (423, 0, 247)]],
}


def light_radiance():
    """main.cc:35: 8*(0.747+0.058, 0.747+0.258, 0.747) + 15.6*(...) + 18.4*(...), in fp32."""
    def v(a, b, c):
        return np.array([f32(a), f32(b), f32(c)], np.float32)
    r = (f32(8.0) * v(f32(0.747) + f32(0.058), f32(0.747) + f32(0.258), 0.747)
         + f32(15.6) * v(f32(0.740) + f32(0.287), f32(0.740) + f32(0.160), 0.740)
         + f32(18.4) * v(f32(0.737) + f32(0.642), f32(0.737) + f32(0.159), 0.737))
    return r.astype(np.float32)


def _normalize(v):
    v = np.asarray(v, np.float32)
    l2 = f32(v[0] * v[0]) + f32(v[1] * v[1])
    l2 = f32(l2) + f32(v[2] * v[2])
    ln = np.sqrt(f32(l2), dtype=np.float32)
    return (v / ln).astype(np.float32)


def cornell_assets():
    d = asset_dir()
    out = {}
    for name, quads in CORNELL.items():
        p = os.path.join(d, "cornell_%s.obj" % name)
        if not os.path.exists(p):
            quads_to_obj(p, quads)
        out[name] = p
    return out


def bunny_mesh(n_lon=187, n_lat=188):
    """Deterministic bunny-shaped stand-in for scene\\bunny\\bunny.obj (not in the reference repository): a closed
    displaced sphere (body + head + two ears) in the bunny's native scale -- about 0.15 units tall, resting
    near y = 0.03 -- with 2*n_lon*(n_lat-1) triangles (69,938 at the defaults)."""
    th = (np.arange(1, n_lat, dtype=np.float64) / n_lat) * np.pi          # interior latitudes
    ph = (np.arange(n_lon, dtype=np.float64) / n_lon) * 2.0 * np.pi
    T, P = np.meshgrid(th, ph, indexing="ij")

    def dirs(t, p):
        return np.stack([np.sin(t) * np.cos(p), np.cos(t), np.sin(t) * np.sin(p)], -1)

    def radius(d):
        r = 0.055 + 0.0 * d[..., 0]
        lobes = [((0.55, 0.80, 0.0), 0.030, 10.0),      # head
                 ((0.45, 0.95, 0.22), 0.060, 60.0),     # ear
                 ((0.45, 0.95, -0.22), 0.060, 60.0),    # ear
                 ((-0.9, -0.2, 0.0), 0.020, 12.0),      # tail
                 ((0.0, -1.0, 0.0), -0.012, 3.0)]       # flattened base
        for c, amp, sharp in lobes:
            c = np.array(c) / np.linalg.norm(c)
            r = r + amp * np.exp(sharp * (d @ c - 1.0))
        r = r + 0.0015 * np.sin(14 * d[..., 0] * np.pi) * np.sin(11 * d[..., 2] * np.pi) * np.sin(9 * d[..., 1] * np.pi)  # fur-scale detail
        return r

    D = dirs(T, P)
    V = D * radius(D)[..., None]
    top = np.array([0.0, 1.0, 0.0]); bot = np.array([0.0, -1.0, 0.0])
    vt = top * radius(top[None])[0]; vb = bot * radius(bot[None])[0]
    verts = np.concatenate([V.reshape(-1, 3), vt[None], vb[None]], 0)
    verts[:, 1] += 0.1                                                  # centre at y = 0.1
    nring = n_lat - 1
    it, ib = nring * n_lon, nring * n_lon + 1
    faces = []
    idx = lambda i, j: i * n_lon + (j % n_lon)
    for j in range(n_lon):
        faces.append((it, idx(0, j + 1), idx(0, j)))
        faces.append((ib, idx(nring - 1, j), idx(nring - 1, j + 1)))
    i = np.arange(nring - 1)[:, None]; j = np.arange(n_lon)[None, :]
    a = i * n_lon + j; b = i * n_lon + (j + 1) % n_lon; c = (i + 1) * n_lon + j; e = (i + 1) * n_lon + (j + 1) % n_lon
    quads = np.stack([np.stack([a, b, e], -1), np.stack([a, e, c], -1)], -2).reshape(-1, 3)
    faces = np.concatenate([np.array(faces, np.int64), quads], 0)
    return verts.astype(np.float32), faces


def bunny_asset(n_lon=187, n_lat=188):
    p = os.path.join(asset_dir(), "bunny_%dx%d.obj" % (n_lon, n_lat))
    if not os.path.exists(p):
        v, f = bunny_mesh(n_lon, n_lat)
        write_obj(p, v, f)
    return p


def file_sha1(path):
    h = hashlib.sha1()
    with open(path, "rb") as fh:
        h.update(fh.read())
    return h.hexdigest()


# ---- backend over the product's host library ------------------------------------------------------------------
class HostBackend:
    prefix = "jp_host_"

    def __init__(self, name="scene", lib=None):
        if lib is None:
            from . import host_lib
            lib = host_lib()
        self.L = lib
        self._new(name)

    def _new(self, name):
        self.h = C.c_void_p(self.L.jp_host_scene_new(name.encode()))

    def _f(self, n):
        return getattr(self.L, self.prefix + n)

    def camera(self, lookfrom, front, up, vfov, resx, resy):
        a, _a = _fa(lookfrom); b, _b = _fa(front); c, _c = _fa(up)
        self._f("scene_camera")(self.h, a, b, c, C.c_float(vfov), C.c_float(resx), C.c_float(resy))

    def envlight(self, rgb):
        a, _a = _fa(rgb); return self._f("scene_envlight")(self.h, a)

    def pointlight(self, pos, intensity):
        a, _a = _fa(pos); b, _b = _fa(intensity); return self._f("scene_pointlight")(self.h, a, b)

    def dirlight(self, direction, irradiance):
        a, _a = _fa(direction); b, _b = _fa(irradiance); return self._f("scene_dirlight")(self.h, a, b)

    def mat_matte(self, rgb):
        a, _a = _fa(rgb); return self._f("mat_matte")(self.h, a)

    def mat_mirror(self, rgb):
        a, _a = _fa(rgb); return self._f("mat_mirror")(self.h, a)

    def mat_glass(self, eta, kr, kt):
        a, _a = _fa(kr); b, _b = _fa(kt); return self._f("mat_glass")(self.h, C.c_float(eta), a, b)

    def mat_plastic(self, kd, ks, rough, remap=False):
        a, _a = _fa(kd); b, _b = _fa(ks); return self._f("mat_plastic")(self.h, a, b, C.c_float(rough), int(remap))

    def mat_metal(self, eta, k, ur, vr, remap=False):
        a, _a = _fa(eta); b, _b = _fa(k); return self._f("mat_metal")(self.h, a, b, C.c_float(ur), C.c_float(vr), int(remap))

    def mesh(self, path, flip_normal, flip_hand, offset=(0, 0, 0), scale=1.0, mat=-1, radiance=None):
        o, _o = _fa(offset)
        r, _r = _fa(radiance) if radiance is not None else (None, None)
        return self._f("scene_mesh")(self.h, path.encode(), int(flip_normal), int(flip_hand), o, C.c_float(scale), mat, r)

    def rect(self, axis, a0, a1, b0, b1, c, flip=False, mat=-1, radiance=None):
        r, _r = _fa(radiance) if radiance is not None else (None, None)
        self._f("scene_rect")(self.h, axis, C.c_float(a0), C.c_float(a1), C.c_float(b0), C.c_float(b1), C.c_float(c), int(flip), mat, r)

    def sphere(self, center, radius, mat=-1, radiance=None):
        cc, _c = _fa(center)
        r, _r = _fa(radiance) if radiance is not None else (None, None)
        self._f("scene_sphere")(self.h, cc, C.c_float(radius), mat, r)

    def disk(self, pos, normal, radius, mat=-1, radiance=None):
        pp, _p = _fa(pos); nn, _n = _fa(normal)
        r, _r = _fa(radiance) if radiance is not None else (None, None)
        self._f("scene_disk")(self.h, pp, nn, C.c_float(radius), mat, r)

    def preprocess(self):
        self._f("scene_preprocess")(self.h)

    def set_reference_tree(self, on=True, certified=False):
        """FScene::referenceTree: build the reference's own tree (its rand() sequence, its std::sort) and have the device
        walk it with the reference's semantics -- bit-identical hits on meshes, several times slower (host backend only).
        certified=True (FScene::certifiedWalk): the certified walk over that tree's leaves, about twice as fast; proven identical ray
        by ray except for secondary rays within fp32 noise of a triangle's plane (about one in 10^9; a generous cull slack reaches those seen: the 280k-triangle frames measured are bit-identical)."""
        self._f("scene_set_reference_tree")(self.h, (2 if certified else 1) if on else 0)

    def set_device_build(self, on=True):
        """FScene::deviceBuild: leave the hierarchy to jp_upload_scene's device LBVH pass (host backend only)."""
        self._f("scene_set_device_build")(self.h, 1 if on else 0)

    def num_primitives(self):
        return self._f("num_primitives")(self.h)

    def num_lights(self):
        return self._f("num_lights")(self.h)

    # host-only
    def flatten(self):
        p = self.L.jp_host_flatten(self.h)
        if not p:
            raise RuntimeError("flatten failed: %s" % self.L.jp_host_last_error(self.h).decode())
        return p

    def close(self):
        if self.h:
            self._f("scene_free")(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


AXIS_XY, AXIS_XZ, AXIS_YZ = 0, 1, 2


# ---- scene scripts (the calls of main.cc) -----------------------------------------------------------------------
def build_cornell(be, width, height, lambert_only=False, extras=None, env=(0.0, 0.0, 0.0)):
    """create_cornellbox_scene main.cc:13-62.  lambert_only swaps the metal tall box for white matte (config C2)."""
    lookfrom = np.array([278, 273, 960], np.float32); lookat = np.array([278, 273, 0], np.float32)
    be.camera(lookfrom, _normalize(lookat - lookfrom), (0, 1, 0), 60.0, width, height)
    be.envlight(env)
    red = be.mat_matte((0.63, 0.065, 0.05))
    green = be.mat_matte((0.14, 0.45, 0.091))
    white = be.mat_matte((0.725, 0.71, 0.68))
    golden = be.mat_metal((0.18, 0.15, 0.81), (0.11, 0.11, 0.11), 0.2, 0.2, False)
    mat_light = be.mat_matte((0.65, 0.65, 0.65))
    A = cornell_assets()
    be.mesh(A["light"], True, True, mat=mat_light, radiance=light_radiance())
    be.mesh(A["floor"], True, True, mat=white)
    be.mesh(A["shortbox"], True, True, mat=white)
    be.mesh(A["tallbox"], True, True, mat=(white if lambert_only else golden))
    be.mesh(A["left"], True, True, mat=red)
    be.mesh(A["right"], True, True, mat=green)
    if extras:
        extras(be, dict(red=red, green=green, white=white, golden=golden))
    be.preprocess()
    return be


def build_bunny(be, width, height, n_lon=187, n_lat=188, instances=4, obj_path=None):
    """create_bunny_scene main.cc:64-111 with the procedural stand-in mesh (4 instances as in the reference)."""
    lookfrom = np.array([-300, 300, -300], np.float32); lookat = np.array([0, 0, 0], np.float32)
    be.camera(lookfrom, _normalize(lookat - lookfrom), (0, 1, 0), 60.0, width, height)
    be.envlight((0.1, 0.1, 0.5))
    red = be.mat_matte((0.63, 0.065, 0.05))
    green = be.mat_matte((0.14, 0.45, 0.091))
    be.mat_matte((0.725, 0.71, 0.68))
    mat_light = be.mat_matte((0.65, 0.65, 0.65))
    be.rect(AXIS_XZ, -100, 100, -100, 100, 350, True, mat_light, light_radiance())
    be.rect(AXIS_XZ, -200, 200, -200, 200, 0, False, green, None)
    obj = obj_path or bunny_asset(n_lon, n_lat)
    kd = np.array([0.35, 0.12, 0.48], np.float32)
    mats = [lambda: red,
            lambda: be.mat_plastic(kd, (np.float32(1) - kd).astype(np.float32), 0.1, False),
            lambda: be.mat_metal((0.18, 0.15, 0.81), (0.11, 0.11, 0.11), 0.2, 0.2, False),
            lambda: be.mat_glass(1.5, (0.98, 0.98, 0.98), (0.98, 0.98, 0.98))]
    offsets = [(0, 0, 0), (-100, 0, -100), (0, 0, -100), (-100, 0, 0)]
    for i in range(instances):
        be.mesh(obj, True, True, offsets[i], 500.0, mats[i](), None)
    be.preprocess()
    return be


def build_misc(be, width, height):
    """Coverage scene for API paths no benchmark scene reaches: FSphere as primitive and as area light
    (shape.h:476-662), FMirrorMaterial, glass sphere (the commented-out one of main.cc:56-58), remapped
    roughness, a null-material primitive (integrator.cc:349-353) and a non-black environment."""
    def extras(b, m):
        glass = b.mat_glass(1.5, (0.98, 0.98, 0.98), (0.98, 0.98, 0.98))
        mirror = b.mat_mirror((0.9, 0.9, 0.9))
        plastic = b.mat_plastic((0.35, 0.12, 0.48), (0.3, 0.3, 0.3), 0.3, True)
        b.sphere((273, 273, 150), 60.0, glass, None)
        b.sphere((120, 330, 300), 40.0, mirror, None)
        b.sphere((420, 90, 120), 50.0, plastic, None)
        b.sphere((278, 440, 280), 25.0, m["white"], (np.array([20, 16, 12], np.float32)))
        b.rect(AXIS_XY, 150, 400, 100, 400, 500, False, -1, None)          # null material: rays pass through
    be_ = build_cornell(be, width, height, lambert_only=False, extras=extras, env=(0.05, 0.08, 0.2))
    return be_


def build_lights(be, width, height):
    """Cornell box lit additionally by the delta lights of light.h:81-180 (FPointLight, FDirectionLight -- only in
    commented-out lines of main.cc:38,82) and a dim environment."""
    def extras(b, m):
        b.pointlight((278, 273, -200), (630000.0 * 0.2, 650000.0 * 0.2, 650000.0 * 0.2))
        b.dirlight((0.3, -1.0, -0.6), (1.5, 1.2, 0.9))
    return build_cornell(be, width, height, lambert_only=False, extras=extras, env=(0.02, 0.02, 0.05))


def build_disks(be, width, height):
    """Cornell box with FDisk shapes (shape.h:189-275; no configuration of main.cc creates one): a disk area light below the
    ceiling, tilted matte / metal / glass disks as occluders, a disk parallel to an axis-aligned wall, a tiny and a large one."""
    def extras(b, m):
        b.disk((278, 540, -279.5), (0.0, -1.0, 0.0), 60.0, m["white"], (17.0, 12.0, 4.0))       # a second, round light, facing down
        b.disk((150, 120, -200), (0.3, 0.8, 0.5), 70.0, m["white"], None)
        metal = b.mat_metal((0.2, 0.9, 1.1), (3.9, 2.4, 2.2), 0.15, 0.3, False)
        b.disk((400, 200, -350), (-0.6, 0.5, 0.7), 90.0, metal, None)
        glass = b.mat_glass(1.5, (0.98, 0.98, 0.98), (0.98, 0.98, 0.98))
        b.disk((278, 300, -150), (0.1, 0.2, 1.0), 55.0, glass, None)
        b.disk((1.0, 274, -280), (1.0, 0.0, 0.0), 100.0, m["white"], None)                        # just in front of the left wall
        b.disk((300, 1.5, -100), (0.0, 1.0, 0.0), 3.0, metal, None)
    return build_cornell(be, width, height, lambert_only=False, extras=extras, env=(0.03, 0.03, 0.03))


def export_reference_layout(root, n_lon=187, n_lat=188):
    """Writes the synthetic assets in the directory layout main.cc expects (scene/cornellbox/*.obj, scene/bunny/bunny.obj)
    for the command-line front end (jet-pbrt_amd/host/jetpbrt --assets ROOT)."""
    os.makedirs(os.path.join(root, "cornellbox"), exist_ok=True)
    os.makedirs(os.path.join(root, "bunny"), exist_ok=True)
    for name, quads in CORNELL.items():
        quads_to_obj(os.path.join(root, "cornellbox", name + ".obj"), quads)
    v, f = bunny_mesh(n_lon, n_lat)
    write_obj(os.path.join(root, "bunny", "bunny.obj"), v, f)
    return root


if __name__ == "__main__":
    import sys
    print(export_reference_layout(sys.argv[1] if len(sys.argv) > 1 else "scene"))
