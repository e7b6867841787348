#!/usr/bin/env python3
"""Generates the golden fixtures (primary hits, Perlin samples, whole paths) under tests/golden/ with oracle/pyoracle.py.

    python tests/golden/make_golden.py [scene ...]

The reference holds no golden vectors of its own (it has no tests), so these are produced by
the independent Python restatement of its intersection code.  Each fixture stores the rays,
the hit records pyoracle computed for them, and — for the Perlin fixture — tables and samples.
Runs on the CPU in this repository only; nothing here reads the reference checkout.
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import golden_scenes as G  # noqa: E402
import pyoracle as P  # noqa: E402


def make_rays(cam_args, n):
    cam = P.Camera(cam_args["look_from"], cam_args["look_at"], cam_args["vup"], cam_args["vfov"], cam_args["aperture"],
                   cam_args["focus_dist"], cam_args["aspect"])
    rays = np.zeros((n * n, 7), dtype=np.float32)
    for j in range(n):
        for i in range(n):
            u = (np.float32(i) + np.float32(0.5)) / np.float32(n - 1)
            v = (np.float32(j) + np.float32(0.5)) / np.float32(n - 1)
            r = cam.ray_through(u, v, time=np.float32(((i * 7 + j * 3) % 10) / 10.0))
            rays[j * n + i] = (*r.o, *r.d, r.tm)
    return rays


def make_scene(name, intended=False):
    """intended: the labelled alternative semantics (RR_SEMANTICS_INTENDED) -> hits_<name>_intended.npz."""
    P.INTENDED = intended
    fn, seed, n = G.SCENES[name]
    A = G.PyAdapter(seed)
    objs, cam = fn(A)
    n_prims = P.number_primitives(objs)
    rays = make_rays(cam, n)
    out = np.zeros(len(rays), dtype=[("prim_id", "<i4"), ("t", "<f4"), ("point", "<f4", 3), ("normal", "<f4", 3),
                                     ("u", "<f4"), ("v", "<f4"), ("front_face", "<u4")])
    t0 = time.time()
    for i, q in enumerate(rays):
        r = P.Ray(tuple(q[0:3]), tuple(q[3:6]), q[6])
        h = P.scene_intersect(objs, r, rnd=P.PathStream(G.MEDIA_SEED, i, 0, 0))
        if h is None:
            out[i]["prim_id"] = -1
        else:
            out[i] = (h.prim, h.t, h.p, h.n, h.u, h.v, 1 if h.front else 0)
    hits = int((out["prim_id"] >= 0).sum())
    print(f"{name}: {len(rays)} rays, {hits} hits, {n_prims} prims, {len(set(out['prim_id']))} distinct ids, {time.time() - t0:.1f} s")
    np.savez_compressed(os.path.join(HERE, f"hits_{name}{'_intended' if intended else ''}.npz"), rays=rays, hits=out, n_prims=n_prims,
                        media_seed=G.MEDIA_SEED)
    P.INTENDED = False


def make_perlin():
    rng = P.PCG32(21)
    vec, perms = P.perlin_tables(rng)
    rs = np.random.RandomState(2)
    pts = np.concatenate([rs.uniform(-300, 300, (96, 3)), rs.uniform(-3, 3, (32, 3))]).astype(np.float32)
    noise = np.array([P.perlin_noise(vec, perms, p) for p in pts], dtype=np.float32)
    turb = np.array([P.perlin_turb(vec, perms, p, 7) for p in pts], dtype=np.float32)
    np.savez_compressed(os.path.join(HERE, "perlin.npz"), seed=21, ranvec=vec, perms=perms, points=pts, noise=noise, turb=turb)
    print("perlin: 128 points")


def make_paths():
    """Whole paths: the radiance of every sample of a small render of tests/golden_paths.py's scene (all material and
    texture kinds, a medium, a moving sphere, a thin lens), by pyoracle's restatement of ray_color."""
    import golden_paths as GP
    t0 = time.time()
    samples = GP.render_py()
    nolight = GP.render_py(light=False)  # empty `lights`: the 0/0 = NaN samples of render.rs:34
    np.savez_compressed(os.path.join(HERE, "paths_box.npz"), samples=samples, samples_nolight=nolight,
                        **{k: np.asarray(v) for k, v in GP.RENDER.items()})
    print(f"paths_box: {samples.shape[0] * samples.shape[1] * samples.shape[2]} samples, mean {np.nanmean(samples):.4f}; without a light "
          f"{np.isnan(nolight).any(axis=-1).mean():.2f} of the samples are NaN; {time.time() - t0:.1f} s")


def make_paths_book():
    """The same scene under the labelled book semantics (RR_SEMANTICS_BOOK): specular materials keep their own ray, `lights`
    is a list (golden_paths.LIGHT_LIST), the traversal is the intended one."""
    import golden_paths as GP
    t0 = time.time()
    samples = GP.render_py(book=True)
    np.savez_compressed(os.path.join(HERE, "paths_box_book.npz"), samples=samples, **{k: np.asarray(v) for k, v in GP.RENDER.items()})
    print(f"paths_box_book: {samples.shape[0] * samples.shape[1] * samples.shape[2]} samples, mean {np.nanmean(samples):.4f}, "
          f"{np.isnan(samples).any(axis=-1).mean():.3f} NaN; {time.time() - t0:.1f} s")


if __name__ == "__main__":
    names = sys.argv[1:] or list(G.SCENES) + [n + "_intended" for n in G.INTENDED_SCENES] + ["perlin", "paths", "paths_book"]
    for nm in names:
        if nm == "perlin":
            make_perlin()
        elif nm == "paths":
            make_paths()
        elif nm == "paths_book":
            make_paths_book()
        elif nm.endswith("_intended"):
            make_scene(nm[:-len("_intended")], intended=True)
        else:
            make_scene(nm)
