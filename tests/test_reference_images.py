"""The one pin this build did not author: the Cornell-box images the reference repository itself holds.

`output/mixture_pdf.png` and `output/cornell_box.png` of the reference are 600x600 renders of cornell_box()
(src/main.rs:325-368) through ray_color's mixture-PDF estimator (src/render.rs:15-41, src/pdf.rs:33-53) and the
per-slice sqrt tonemap of render_worker (src/render.rs:102-119).  Their spp, thread count and seeds are unknown and the
reference's RNG is unseeded, so the comparison is statistical: 8-bit image against 8-bit image after a 4x4 box filter
(which removes most of the per-pixel Monte-Carlo noise of both sides).

What a pass pins, independent of revision details: the scene geometry and camera (walls, boxes, their rotation and
placement land on the same pixels: correlation > 0.995), the estimator (emitted + albedo * scattering_pdf * Li / mixture
pdf with the XZRect light sampled half of the time: the two neighbouring images of the SAME scene rendered with the
cosine-only estimator and with direct light sampling are 6x and 15x farther away and must stay farther), the light's
power and the Lambertian albedos (channel means within 3 %), and the tonemap (sqrt gamma, 256 scale, clamp).
What it cannot pin: anything the Cornell box does not contain (BVH traversal quirks, triangles, media, textures, moving
spheres, specular materials) and last-bit arithmetic.

Tests marked `gpu` render on the device through the C ABI; the others run the C++ oracle.
"""
import os

import numpy as np
import pytest

import oracle_lib as O

REF_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_output")
MATCH = ("mixture_pdf", "cornell_box")
OTHER_ESTIMATORS = ("cosin_density_pdf", "light_sampled_directly")

RMSE_MAX = 0.02          # of 255, after the box filter; measured 0.007-0.013 (oracle 256-1024 spp), 0.006-0.008 (device 5000 spp)
MEAN_REL_MAX = 0.03      # per channel
CORR_MIN = 0.995


def load_png(name):
    from PIL import Image
    return np.asarray(Image.open(os.path.join(REF_DIR, name + ".png")).convert("RGB")).astype(np.float64)


def box_filter(img, f=4):
    h, w, c = img.shape
    return img[:h - h % f, :w - w % f].reshape(h // f, f, w // f, f, c).mean(axis=(1, 3))


def distance(a, b):
    """(RMSE / 255 of the box-filtered images, worst relative difference of the channel means, correlation)."""
    fa, fb = box_filter(a), box_filter(b)
    rmse = float(np.sqrt(((fa - fb) ** 2).mean()) / 255.0)
    ma, mb = a.mean(axis=(0, 1)), b.mean(axis=(0, 1))
    mean_rel = float(np.abs(ma / mb - 1.0).max())
    corr = float(np.corrcoef(fa.ravel(), fb.ravel())[0, 1])
    return rmse, mean_rel, corr


def check_against_reference_images(rgb8, label):
    img = rgb8.astype(np.float64)
    assert img.shape == (600, 600, 3)
    d = {n: distance(img, load_png(n)) for n in MATCH + OTHER_ESTIMATORS}
    for n, (rmse, mean_rel, corr) in d.items():
        print(f"{label} vs {n}.png: rmse/255 {rmse:.4f}, channel means within {100 * mean_rel:.2f} %, correlation {corr:.4f}")
    for n in MATCH:
        rmse, mean_rel, corr = d[n]
        assert rmse <= RMSE_MAX, (n, rmse)
        assert mean_rel <= MEAN_REL_MAX, (n, mean_rel)
        assert corr >= CORR_MIN, (n, corr)
        for o in OTHER_ESTIMATORS:  # the same scene through another estimator must be clearly farther away
            assert d[o][0] > 3.0 * rmse, (n, o, d[o][0], rmse)
    return d


def test_the_reference_images_agree_with_each_other():
    """The yardstick: two runs of the reference on the same scene differ by this much (unknown spp each)."""
    rmse, mean_rel, corr = distance(load_png(MATCH[0]), load_png(MATCH[1]))
    print(f"mixture_pdf.png vs cornell_box.png: rmse/255 {rmse:.4f}, means {100 * mean_rel:.2f} %, correlation {corr:.4f}")
    assert rmse <= RMSE_MAX and mean_rel <= MEAN_REL_MAX and corr >= CORR_MIN


@pytest.mark.parametrize("n_slices,spp", [(8, 512), (16, 768)])
def test_oracle_cornell_box_matches_the_images_the_reference_holds(built, n_slices, spp):
    """C++ oracle, cornell_box() as main.rs has it, 600x600, depth 200, 64 and 48 samples per slice (the reference's own run
    has 5000 / n_threads per slice, main.rs:331; the mean of square roots of render.rs:102-119 sits ~1 % lower at 32, where
    the blue channel is 3.0 % under mixture_pdf.png; the two reference images differ by 2.3 % from each other)."""
    from rust_raytracer_amd import host
    b, flat, d = host.build_config_scene("cornell")
    assert (d["width"], d["height"], d["max_depth"]) == (600, 600, 200)
    p = O.make_params(600, 600, spp, n_slices=n_slices, max_depth=200, seed=1)
    r = O.render(flat, p, n_threads=min(16, len(os.sched_getaffinity(0))))
    check_against_reference_images(r["rgb8"], f"oracle K={n_slices} {spp} spp")


@pytest.mark.gpu
@pytest.mark.parametrize("n_slices,spp", [(8, 1000), (16, 5000)])
def test_device_cornell_box_matches_the_images_the_reference_holds(built, gpu_ctx, n_slices, spp):
    """The HIP path through the C ABI: configs[1] (1000 spp) and the run main.rs itself asks for (5000 spp)."""
    from rust_raytracer_amd import device, host
    b, flat, d = host.build_config_scene("cornell")
    gpu_ctx.upload(flat)
    p = device.make_params(600, 600, spp, n_slices=n_slices, max_depth=200, seed=1)
    r = gpu_ctx.render(p, slice_sums=False)
    dist = check_against_reference_images(r["rgb8"], f"device K={n_slices} {spp} spp")
    for n in MATCH:  # at these sample counts the device is as close to a reference run as the two reference runs are to each other
        assert dist[n][0] <= 0.012, (n, dist[n])


# ------------------------------------------------------------------------------------------ the labelled book semantics
def _book_scene(name, **kw):
    from rust_raytracer_amd import abi, host
    b, flat, d = host.build_config_scene(name, **kw)
    flat.semantics = abi.SEMANTICS_BOOK
    return b, flat, d


@pytest.mark.gpu
def test_book_semantics_render_the_other_images_the_reference_holds(built, gpu_ctx):
    """`output/suzanne.png` and `output/final_scene.png` of the reference were rendered by earlier revisions: the light is
    part of the scene, metal, glass and smoke carry light, the whole BVH is visible.  HEAD renders neither (black: the light
    is not in the scene, main.rs:451,513-514; and with it added, every non-Lambertian surface stays black, material.rs:170-175).
    RR_SEMANTICS_BOOK (never the default: intended traversal + the book's estimator) on the scenes of main.rs:416-464 and
    :466-541 with the light added renders what final_scene.png shows.  A weak pin: the random box heights and the random
    sphere cluster differ from the run that made the image, so the thresholds are loose — and still far from what the
    reference's own semantics give on the same scene."""
    from rust_raytracer_amd import abi, device
    results = {}
    for name, png, side, depth, kw in (("suzanne", "suzanne", 600, 100, {}), ("final", "final_scene", 800, 50, dict(moving_sphere=False))):
        ref = load_png(png)
        for sem in ("book", "reference"):
            b, flat, d = _book_scene(name, **kw)
            if sem == "reference":
                flat.semantics = abi.SEMANTICS_REFERENCE
            gpu_ctx.upload(flat)
            r = gpu_ctx.render(device.make_params(side, side, 512, n_slices=8, max_depth=depth, seed=3), slice_sums=False)
            results[(name, sem)] = distance(r["rgb8"].astype(np.float64), ref)
            rmse, mean_rel, corr = results[(name, sem)]
            print(f"{name} under {sem} semantics vs {png}.png: rmse/255 {rmse:.4f}, channel means within {100 * mean_rel:.1f} %, correlation {corr:.4f}")
    # final_scene.png (measured: rmse/255 0.105, channel means within 1.3 %, correlation 0.907; under the reference's own
    # semantics 0.218, 51 %, 0.804: glass, metal, smoke and most of the BVHs are missing there)
    rmse, mean_rel, corr = results[("final", "book")]
    assert rmse <= 0.13 and mean_rel <= 0.05 and corr >= 0.88
    assert results[("final", "reference")][0] > 1.7 * rmse and results[("final", "reference")][1] > 0.3
    # suzanne.png pins no more than the camera and the room: the revision that made it had a far brighter light than the
    # (2, 2, 2) of main.rs:447 (its walls are saturated: channel means 55 % above any render of the scene as HEAD has it) and
    # a reflecting Suzanne where HEAD's is white Lambertian (main.rs:459), so neither semantics is nearer to it than the other
    assert results[("suzanne", "book")][2] >= 0.75 and results[("suzanne", "reference")][2] >= 0.75
