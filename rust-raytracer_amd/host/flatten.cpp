// Flattening of the host Geometry tree into the tagged arrays of rr_flat_scene.
// The reference has no such step (it traverses Box<Geometry> directly,
// src/collider.rs:682-697); this is the host half of the device boundary.
#include <unordered_map>

#include "scene.hpp"

namespace rr {

namespace {

struct Flattener {
    FlatScene& out;
    const SceneColliders& scene;
    std::unordered_map<const MaterialData*, uint32_t> material_ids;
    std::unordered_map<const TextureData*, uint32_t> texture_ids;
    std::unordered_map<const Perlin*, uint32_t> perlin_ids;

    uint32_t texture_id(const Texture& t) {
        const TextureData* d = t.data().get();
        if (!d) throw std::invalid_argument("flatten: material without a texture");
        auto it = texture_ids.find(d);
        if (it != texture_ids.end()) return it->second;
        rr_texture r{};
        r.type = d->type;
        r.color0[0] = d->color0.x;
        r.color0[1] = d->color0.y;
        r.color0[2] = d->color0.z;
        r.color1[0] = d->color1.x;
        r.color1[1] = d->color1.y;
        r.color1[2] = d->color1.z;
        r.scale = d->scale;
        r.depth = d->depth;
        if (d->type == RR_TEX_NOISE) {
            const Perlin* p = d->perlin.get();
            auto pit = perlin_ids.find(p);
            if (pit == perlin_ids.end()) {
                rr_perlin rp;
                static_assert(sizeof(rp.ranvec) == sizeof(p->ranvec), "perlin layout");
                std::copy(&p->ranvec[0][0], &p->ranvec[0][0] + RR_PERLIN_POINTS * 3, &rp.ranvec[0][0]);
                std::copy(p->perm_x, p->perm_x + RR_PERLIN_POINTS, rp.perm_x);
                std::copy(p->perm_y, p->perm_y + RR_PERLIN_POINTS, rp.perm_y);
                std::copy(p->perm_z, p->perm_z + RR_PERLIN_POINTS, rp.perm_z);
                out.perlin.push_back(rp);
                pit = perlin_ids.emplace(p, static_cast<uint32_t>(out.perlin.size() - 1)).first;
            }
            r.index = pit->second;
        } else if (d->type == RR_TEX_IMAGE) {
            if (d->image >= scene.atlas.size()) throw std::invalid_argument("flatten: image texture index outside the atlas");
            r.index = d->image;
        }
        out.textures.push_back(r);
        uint32_t id = static_cast<uint32_t>(out.textures.size() - 1);
        texture_ids.emplace(d, id);
        return id;
    }

    uint32_t material_id(const Material& m) {
        const MaterialData* d = m.data().get();
        if (!d) throw std::invalid_argument("flatten: primitive without a material");
        auto it = material_ids.find(d);
        if (it != material_ids.end()) return it->second;
        rr_material r{};
        r.type = d->type;
        r.texture = texture_id(d->texture);
        r.color[0] = d->color.x;
        r.color[1] = d->color.y;
        r.color[2] = d->color.z;
        r.param = d->param;
        out.materials.push_back(r);
        uint32_t id = static_cast<uint32_t>(out.materials.size() - 1);
        material_ids.emplace(d, id);
        return id;
    }

    uint32_t add_rect(const Geometry& g) {
        rr_rect r{};
        r.a0 = g.f[0];
        r.a1 = g.f[1];
        r.b0 = g.f[2];
        r.b1 = g.f[3];
        r.k = g.f[4];
        r.axis = g.type == GeometryType::XYRect ? RR_RECT_XY : (g.type == GeometryType::XZRect ? RR_RECT_XZ : RR_RECT_YZ);
        r.material = material_id(g.material);
        r.prim_id = out.n_prims++;
        out.rects.push_back(r);
        return static_cast<uint32_t>(out.rects.size() - 1);
    }

    uint32_t add_list(const std::vector<GeometryPtr>& items) {
        // Reserve the list record first so that a parent precedes its children.
        const uint32_t li = static_cast<uint32_t>(out.lists.size());
        out.lists.push_back(rr_list{0, 0});
        std::vector<uint32_t> refs;
        refs.reserve(items.size());
        for (const auto& it : items) refs.push_back(visit(*it));
        out.lists[li].first_child = static_cast<uint32_t>(out.child_refs.size());
        out.lists[li].n_children = static_cast<uint32_t>(refs.size());
        out.child_refs.insert(out.child_refs.end(), refs.begin(), refs.end());
        return li;
    }

    uint32_t visit(const Geometry& g) {
        switch (g.type) {
            case GeometryType::Sphere: {
                rr_sphere r{};
                r.center[0] = g.p[0].x;
                r.center[1] = g.p[0].y;
                r.center[2] = g.p[0].z;
                r.radius = g.f[0];
                r.material = material_id(g.material);
                r.prim_id = out.n_prims++;
                out.spheres.push_back(r);
                return RR_REF(RR_TAG_SPHERE, out.spheres.size() - 1);
            }
            case GeometryType::MovingSphere: {
                rr_moving_sphere r{};
                for (int a = 0; a < 3; ++a) {
                    r.center0[a] = g.p[0][a];
                    r.center1[a] = g.p[1][a];
                }
                r.time0 = g.f[0];
                r.time1 = g.f[1];
                r.radius = g.f[2];
                r.material = material_id(g.material);
                r.prim_id = out.n_prims++;
                out.moving_spheres.push_back(r);
                return RR_REF(RR_TAG_MOVING_SPHERE, out.moving_spheres.size() - 1);
            }
            case GeometryType::XYRect:
            case GeometryType::XZRect:
            case GeometryType::YZRect:
                return RR_REF(RR_TAG_RECT, add_rect(g));
            case GeometryType::Triangle: {
                rr_triangle r{};
                for (int a = 0; a < 3; ++a) {
                    r.p0[a] = g.p[0][a];
                    r.p1[a] = g.p[1][a];
                    r.p2[a] = g.p[2][a];
                    r.normal[a] = g.p[3][a];
                }
                r.material = material_id(g.material);
                r.prim_id = out.n_prims++;
                out.triangles.push_back(r);
                return RR_REF(RR_TAG_TRIANGLE, out.triangles.size() - 1);
            }
            case GeometryType::Cuboid: {
                rr_cuboid c{};
                c.first_rect = static_cast<uint32_t>(out.rects.size());
                for (const auto& side : g.children) add_rect(*side);
                out.cuboids.push_back(c);
                return RR_REF(RR_TAG_CUBOID, out.cuboids.size() - 1);
            }
            case GeometryType::ConstantMedium: {
                const uint32_t mi = static_cast<uint32_t>(out.media.size());
                out.media.push_back(rr_medium{});
                rr_medium m{};
                m.material = material_id(g.material);
                m.neg_inv_density = g.f[0];
                m.prim_id = out.n_prims++;
                m.boundary = visit(*g.children[0]);
                out.media[mi] = m;
                return RR_REF(RR_TAG_MEDIUM, mi);
            }
            case GeometryType::TranslateInstance:
            case GeometryType::RotationInstance: {
                const uint32_t ii = static_cast<uint32_t>(out.instances.size());
                out.instances.push_back(rr_instance{});
                rr_instance in{};
                if (g.type == GeometryType::TranslateInstance) {
                    in.kind = RR_INST_TRANSLATE;
                    in.v[0] = g.p[0].x;
                    in.v[1] = g.p[0].y;
                    in.v[2] = g.p[0].z;
                } else {
                    in.kind = g.axis == Axis::X ? RR_INST_ROTATE_X : (g.axis == Axis::Y ? RR_INST_ROTATE_Y : RR_INST_ROTATE_Z);
                    in.v[0] = g.f[1];
                    in.v[1] = g.f[2];
                    in.v[2] = 0.f;
                }
                in.child = visit(*g.children[0]);
                out.instances[ii] = in;
                return RR_REF(RR_TAG_INSTANCE, ii);
            }
            case GeometryType::BVHNode: {
                const uint32_t ni = static_cast<uint32_t>(out.bvh_nodes.size());
                out.bvh_nodes.push_back(rr_bvh_node{});
                rr_bvh_node n{};
                for (int a = 0; a < 3; ++a) {
                    n.bmin[a] = g.has_box ? g.box.minimum[a] : 0.f;
                    n.bmax[a] = g.has_box ? g.box.maximum[a] : 0.f;
                }
                n.axis = g.axis_index;
                n.has_box = g.has_box ? 1u : 0u;
                n.left = visit(*g.children[0]);
                n.right = g.children[1].get() == g.children[0].get() ? n.left : visit(*g.children[1]);
                out.bvh_nodes[ni] = n;
                return RR_REF(RR_TAG_BVH, ni);
            }
            case GeometryType::ColliderList:
                return RR_REF(RR_TAG_LIST, add_list(g.children));
        }
        throw std::logic_error("flatten: unknown geometry type");
    }
};

}  // namespace

std::unique_ptr<FlatScene> flatten(const SceneColliders& scene, const GeometryPtr& lights, const Camera& cam,
                                   Rgb background) {
    auto fs = std::make_unique<FlatScene>();
    Flattener fl{*fs, scene, {}, {}, {}};
    fs->root = RR_REF(RR_TAG_LIST, fl.add_list(scene.objects));

    for (const auto& im : scene.atlas) {
        rr_image r{};
        r.width = im.width;
        r.height = im.height;
        r.offset = fs->image_bytes.size();
        fs->images.push_back(r);
        fs->image_bytes.insert(fs->image_bytes.end(), im.rgb.begin(), im.rgb.end());
    }

    // collider.rs:714-725: only a bare XZRect has pdf_value/random.
    auto as_light = [](const GeometryPtr& g) {
        rr_light l{};
        if (g && g->type == GeometryType::XZRect) {
            l.is_xzrect = 1;
            l.x0 = g->f[0];
            l.x1 = g->f[1];
            l.z0 = g->f[2];
            l.z1 = g->f[3];
            l.k = g->f[4];
        }
        return l;
    };
    fs->light = as_light(lights);
    // (the members of a ColliderList passed as `lights`: the book estimator's light list, unread under the reference's semantics)
    if (lights && lights->type == GeometryType::ColliderList)
        for (const GeometryPtr& member : lights->children) fs->lights.push_back(as_light(member));

    rr_camera& c = fs->camera;
    for (int a = 0; a < 3; ++a) {
        c.origin[a] = cam.origin[a];
        c.lower_left[a] = cam.lower_left[a];
        c.horizontal[a] = cam.horizontal[a];
        c.vertical[a] = cam.vertical[a];
        c.u[a] = cam.u[a];
        c.v[a] = cam.v[a];
    }
    c.lens_radius = cam.lens_radius;
    c.shutter_open = cam.shutter_open;
    c.shutter_close = cam.shutter_close;
    fs->background[0] = background.x;
    fs->background[1] = background.y;
    fs->background[2] = background.z;
    return fs;
}

rr_flat_scene FlatScene::view() const {
    rr_flat_scene v{};
    v.n_spheres = static_cast<uint32_t>(spheres.size());
    v.spheres = spheres.data();
    v.n_moving_spheres = static_cast<uint32_t>(moving_spheres.size());
    v.moving_spheres = moving_spheres.data();
    v.n_rects = static_cast<uint32_t>(rects.size());
    v.rects = rects.data();
    v.n_triangles = static_cast<uint32_t>(triangles.size());
    v.triangles = triangles.data();
    v.n_cuboids = static_cast<uint32_t>(cuboids.size());
    v.cuboids = cuboids.data();
    v.n_media = static_cast<uint32_t>(media.size());
    v.media = media.data();
    v.n_instances = static_cast<uint32_t>(instances.size());
    v.instances = instances.data();
    v.n_bvh_nodes = static_cast<uint32_t>(bvh_nodes.size());
    v.bvh_nodes = bvh_nodes.data();
    v.n_lists = static_cast<uint32_t>(lists.size());
    v.lists = lists.data();
    v.n_child_refs = static_cast<uint32_t>(child_refs.size());
    v.child_refs = child_refs.data();
    v.n_materials = static_cast<uint32_t>(materials.size());
    v.materials = materials.data();
    v.n_textures = static_cast<uint32_t>(textures.size());
    v.textures = textures.data();
    v.n_perlin = static_cast<uint32_t>(perlin.size());
    v.perlin = perlin.data();
    v.n_images = static_cast<uint32_t>(images.size());
    v.images = images.data();
    v.n_image_bytes = image_bytes.size();
    v.image_bytes = image_bytes.data();
    v.root = root;
    v.light = light;
    v.n_lights = static_cast<uint32_t>(lights.size());
    v.lights = lights.empty() ? nullptr : lights.data();
    v.camera = camera;
    v.background[0] = background[0];
    v.background[1] = background[1];
    v.background[2] = background[2];
    return v;
}

}  // namespace rr
