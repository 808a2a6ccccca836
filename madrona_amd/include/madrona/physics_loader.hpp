// Uploads baked collision assets to the GPU and owns the ObjectManager the
// physics systems read.
//
// API contract: reference include/madrona/physics_loader.hpp:11-23 and
// src/physics/physics_loader.cpp (init :38-151, loadRigidBodies :186-333):
// fixed capacity of max_objects objects x 20 primitives, objects appended
// per call, hull pointers rebased onto the device copies.  ExecMode::CUDA is
// the only mode of this backend ("CUDA" = the GPU backend, here HIP); the
// returned ObjectManager lives in device memory: pass its address to the
// simulator, do not dereference it on the host.
#pragma once

#include <madrona/physics.hpp>
#include <madrona/physics_assets.hpp>
#include <madrona/exec_mode.hpp>
#include <madrona/crash.hpp>

#include <mwhip.h>

#include <cstring>
#include <memory>
#include <vector>

namespace madrona::phys {

class PhysicsLoader {
public:
    inline PhysicsLoader(ExecMode exec_mode, CountT max_objects,
                         int gpu_id = 0);
    inline ~PhysicsLoader();
    PhysicsLoader(PhysicsLoader &&o) = default;

    inline CountT loadRigidBodies(const RigidBodyAssets &assets);

    inline ObjectManager &getObjectManager() { return *impl_->mgr; }

private:
    struct Impl {
        CollisionPrimitive *primitives;
        math::AABB *primAABBs;
        math::AABB *objAABBs;
        uint32_t *rigidBodyPrimitiveOffsets;
        uint32_t *rigidBodyPrimitiveCounts;
        RigidBodyMetadata *metadatas;
        ObjectManager *mgr;

        CountT curPrimOffset;
        CountT curObjOffset;
        CountT maxPrims;
        CountT maxObjs;
        int gpuID;
        std::vector<void *> hullAllocs;

        // host mirror of what the first PrimImage::maxPrims primitives need
        // (device-side primitive records + their hull data), and the image
        PrimImage *primImage;
        struct HostPrim {
            CollisionPrimitive devPrim;
            math::AABB aabb;
            std::vector<uint32_t> planes, halfEdges, vertices, faceBase;
        };
        std::vector<HostPrim> hostPrims;
    };

    inline void rebuildPrimImage();

    template <typename T>
    inline T *allocDevice(CountT n)
    {
        void *ptr = mwhip_raw_alloc(impl_->gpuID, sizeof(T) * (uint64_t)n);
        if (ptr == nullptr) {
            FATAL("PhysicsLoader: %s", mwhip_last_error());
        }
        return (T *)ptr;
    }

    inline void upload(void *dst, const void *src, uint64_t num_bytes)
    {
        if (mwhip_raw_copy_h2d(impl_->gpuID, dst, src, num_bytes) != 0) {
            FATAL("PhysicsLoader: %s", mwhip_last_error());
        }
    }

    std::unique_ptr<Impl> impl_;
};

PhysicsLoader::PhysicsLoader(ExecMode exec_mode, CountT max_objects, int gpu_id)
    : impl_(new Impl {})
{
    if (exec_mode != ExecMode::CUDA) {
        FATAL("PhysicsLoader: this backend only has the GPU execution mode");
    }

    constexpr CountT max_prims_per_object = 20;

    impl_->gpuID = gpu_id;
    impl_->maxObjs = max_objects;
    impl_->maxPrims = max_objects * max_prims_per_object;
    impl_->curObjOffset = 0;
    impl_->curPrimOffset = 0;

    impl_->primitives = allocDevice<CollisionPrimitive>(impl_->maxPrims);
    impl_->primAABBs = allocDevice<math::AABB>(impl_->maxPrims);
    impl_->objAABBs = allocDevice<math::AABB>(max_objects);
    impl_->rigidBodyPrimitiveOffsets = allocDevice<uint32_t>(max_objects);
    impl_->rigidBodyPrimitiveCounts = allocDevice<uint32_t>(max_objects);
    impl_->metadatas = allocDevice<RigidBodyMetadata>(max_objects);
    impl_->mgr = allocDevice<ObjectManager>(1);
    impl_->primImage = allocDevice<PrimImage>(1);
    {
        PrimImage empty {};
        upload(impl_->primImage, &empty, sizeof(PrimImage));
    }

    ObjectManager local {
        impl_->primitives,
        impl_->primAABBs,
        impl_->objAABBs,
        impl_->rigidBodyPrimitiveOffsets,
        impl_->rigidBodyPrimitiveCounts,
        impl_->metadatas,
        impl_->primImage,
    };
    upload(impl_->mgr, &local, sizeof(ObjectManager));
}

PhysicsLoader::~PhysicsLoader()
{
    if (impl_ == nullptr) {
        return;
    }

    for (void *ptr : impl_->hullAllocs) {
        mwhip_raw_free(impl_->gpuID, ptr);
    }
    mwhip_raw_free(impl_->gpuID, impl_->primitives);
    mwhip_raw_free(impl_->gpuID, impl_->primAABBs);
    mwhip_raw_free(impl_->gpuID, impl_->objAABBs);
    mwhip_raw_free(impl_->gpuID, impl_->rigidBodyPrimitiveOffsets);
    mwhip_raw_free(impl_->gpuID, impl_->rigidBodyPrimitiveCounts);
    mwhip_raw_free(impl_->gpuID, impl_->metadatas);
    mwhip_raw_free(impl_->gpuID, impl_->mgr);
    mwhip_raw_free(impl_->gpuID, impl_->primImage);
}

// Same placement rules as the per-world staging it replaces
// (phys_impl/world_step.inl stagePrimitives): primitives in index order, a mesh
// shared by several primitives staged once, a mesh that does not fit the arena
// left in HBM.
void PhysicsLoader::rebuildPrimImage()
{
    PrimImage image {};
    const size_t total = (size_t)impl_->curPrimOffset;
    if (total == 0 || total > PrimImage::maxPrims ||
            impl_->hostPrims.size() != total) {
        upload(impl_->primImage, &image, sizeof(PrimImage));
        return;
    }

    image.numPrims = (uint32_t)total;
    uint32_t used = 0;
    for (size_t p = 0; p < total; p++) {
        const Impl::HostPrim &hp = impl_->hostPrims[p];
        image.prims[p] = hp.devPrim;
        image.primAABBs[p] = hp.aabb;
        for (int i = 0; i < 4; i++) image.meshOffset[p][i] = -1;
        if (hp.devPrim.type != CollisionPrimitive::Type::Hull) {
            continue;
        }

        bool shared = false;
        for (size_t q = 0; q < p; q++) {
            const Impl::HostPrim &hq = impl_->hostPrims[q];
            if (hq.devPrim.type == CollisionPrimitive::Type::Hull &&
                    hq.devPrim.hull.halfEdgeMesh.vertices ==
                        hp.devPrim.hull.halfEdgeMesh.vertices) {
                for (int i = 0; i < 4; i++) {
                    image.meshOffset[p][i] = image.meshOffset[q][i];
                }
                shared = true;
                break;
            }
        }
        if (shared) {
            continue;
        }

        const uint32_t need = (uint32_t)(hp.planes.size() +
            hp.halfEdges.size() + hp.vertices.size() + hp.faceBase.size());
        if (used + need > PrimImage::arenaDwords) {
            continue;       // stays in HBM
        }
        uint32_t at = used;
        auto place = [&](const std::vector<uint32_t> &words, int slot) {
            image.meshOffset[p][slot] = (int32_t)at;
            memcpy(image.arena + at, words.data(), words.size() * 4);
            at += (uint32_t)words.size();
        };
        place(hp.planes, 0);
        place(hp.halfEdges, 1);
        place(hp.vertices, 2);
        place(hp.faceBase, 3);
        used = at;
    }
    image.arenaUsed = used;
    upload(impl_->primImage, &image, sizeof(PrimImage));
}

CountT PhysicsLoader::loadRigidBodies(const RigidBodyAssets &assets)
{
    CountT cur_obj_offset = impl_->curObjOffset;
    CountT cur_prim_offset = impl_->curPrimOffset;

    if (cur_obj_offset + (CountT)assets.numObjs > impl_->maxObjs ||
        cur_prim_offset + (CountT)assets.totalNumPrimitives >
            impl_->maxPrims) {
        FATAL("PhysicsLoader: out of object / primitive slots");
    }

    impl_->curObjOffset += assets.numObjs;
    impl_->curPrimOffset += assets.totalNumPrimitives;

    std::vector<uint32_t> offsets(assets.numObjs);
    for (uint32_t i = 0; i < assets.numObjs; i++) {
        offsets[i] = assets.primOffsets[i] + (uint32_t)cur_prim_offset;
    }

    upload(impl_->primAABBs + cur_prim_offset, assets.primitiveAABBs,
           sizeof(math::AABB) * assets.totalNumPrimitives);
    upload(impl_->objAABBs + cur_obj_offset, assets.objAABBs,
           sizeof(math::AABB) * assets.numObjs);
    upload(impl_->rigidBodyPrimitiveOffsets + cur_obj_offset, offsets.data(),
           sizeof(uint32_t) * assets.numObjs);
    upload(impl_->rigidBodyPrimitiveCounts + cur_obj_offset, assets.primCounts,
           sizeof(uint32_t) * assets.numObjs);
    upload(impl_->metadatas + cur_obj_offset, assets.metadatas,
           sizeof(RigidBodyMetadata) * assets.numObjs);

    auto *hull_halfedges =
        allocDevice<geo::HalfEdge>(assets.hullData.numHalfEdges);
    auto *hull_face_base_halfedges =
        allocDevice<uint32_t>(assets.hullData.numFaces);
    auto *hull_face_planes = allocDevice<geo::Plane>(assets.hullData.numFaces);
    auto *hull_verts = allocDevice<math::Vector3>(assets.hullData.numVerts);
    impl_->hullAllocs.push_back(hull_halfedges);
    impl_->hullAllocs.push_back(hull_face_base_halfedges);
    impl_->hullAllocs.push_back(hull_face_planes);
    impl_->hullAllocs.push_back(hull_verts);

    upload(hull_halfedges, assets.hullData.halfEdges,
           sizeof(geo::HalfEdge) * assets.hullData.numHalfEdges);
    upload(hull_face_base_halfedges, assets.hullData.faceBaseHalfEdges,
           sizeof(uint32_t) * assets.hullData.numFaces);
    upload(hull_face_planes, assets.hullData.facePlanes,
           sizeof(geo::Plane) * assets.hullData.numFaces);
    upload(hull_verts, assets.hullData.vertices,
           sizeof(math::Vector3) * assets.hullData.numVerts);

    // rebase the hull pointers of a scratch copy onto the device arrays
    std::vector<CollisionPrimitive> prims(
        assets.primitives, assets.primitives + assets.totalNumPrimitives);
    for (CollisionPrimitive &prim : prims) {
        if (prim.type != CollisionPrimitive::Type::Hull) {
            continue;
        }

        geo::HalfEdgeMesh &he_mesh = prim.hull.halfEdgeMesh;

        CountT hedge_offset = he_mesh.halfEdges - assets.hullData.halfEdges;
        CountT face_offset = he_mesh.facePlanes - assets.hullData.facePlanes;
        CountT vert_offset = he_mesh.vertices - assets.hullData.vertices;

        he_mesh.halfEdges = hull_halfedges + hedge_offset;
        he_mesh.faceBaseHalfEdges = hull_face_base_halfedges + face_offset;
        he_mesh.facePlanes = hull_face_planes + face_offset;
        he_mesh.vertices = hull_verts + vert_offset;
    }

    upload(impl_->primitives + cur_prim_offset, prims.data(),
           sizeof(CollisionPrimitive) * assets.totalNumPrimitives);

    // host mirror for the primitive image (only its first few entries matter)
    for (uint32_t i = 0; i < assets.totalNumPrimitives; i++) {
        if (impl_->hostPrims.size() >= PrimImage::maxPrims + 1) break;
        Impl::HostPrim hp;
        hp.devPrim = prims[i];
        hp.aabb = assets.primitiveAABBs[i];
        if (prims[i].type == CollisionPrimitive::Type::Hull) {
            const geo::HalfEdgeMesh &src = assets.primitives[i].hull.halfEdgeMesh;
            auto words = [](const void *ptr, size_t n) {
                const uint32_t *w = (const uint32_t *)ptr;
                return std::vector<uint32_t>(w, w + n);
            };
            hp.planes = words(src.facePlanes, (size_t)src.numFaces * 4);
            hp.halfEdges = words(src.halfEdges, (size_t)src.numHalfEdges * 3);
            hp.vertices = words(src.vertices, (size_t)src.numVertices * 3);
            hp.faceBase = words(src.faceBaseHalfEdges, (size_t)src.numFaces);
        }
        impl_->hostPrims.push_back(std::move(hp));
    }
    rebuildPrimImage();

    return cur_obj_offset;
}

}
