// ECSRegistry: what the simulator's registerTypes receives.
// API contract: reference include/madrona/registry.hpp:19-76.
#pragma once

#include <madrona/state.hpp>

namespace madrona {

class ECSRegistry {
public:
    // export_ptrs is kept for signature compatibility with the reference; the
    // exported addresses live in the executor (mwhip_get_exported).
    MADRONA_HOST_API inline ECSRegistry(StateManager *state_mgr, void **export_ptrs)
        : state_mgr_(state_mgr), export_ptrs_(export_ptrs)
    {}

    template <typename ComponentT>
    MADRONA_HOST_API void registerComponent(uint32_t num_bytes = 0)
    {
        state_mgr_->registerComponent<ComponentT>(num_bytes);
    }

    template <typename ArchetypeT>
    MADRONA_HOST_API void registerArchetype()
    {
        state_mgr_->registerArchetype<ArchetypeT>(
            ComponentMetadataSelector<> {}, ArchetypeFlags::None, 0);
    }

    template <typename ArchetypeT, typename... MetadataComponentTs>
    MADRONA_HOST_API void registerArchetype(
        ComponentMetadataSelector<MetadataComponentTs...> component_metadatas,
        ArchetypeFlags archetype_flags,
        CountT max_num_entities_per_world = 0)
    {
        state_mgr_->registerArchetype<ArchetypeT>(component_metadatas,
            archetype_flags, max_num_entities_per_world);
    }

    template <typename BundleT>
    MADRONA_HOST_API void registerBundle()
    {
        state_mgr_->registerBundle<BundleT>();
    }

    template <typename AliasT, typename BundleT>
    MADRONA_HOST_API void registerBundleAlias()
    {
        state_mgr_->registerBundleAlias<AliasT, BundleT>();
    }

    template <typename SingletonT>
    MADRONA_HOST_API void registerSingleton()
    {
        state_mgr_->registerSingleton<SingletonT>();
    }

    template <typename ArchetypeT, typename ComponentT>
    MADRONA_HOST_API void exportColumn(int32_t slot)
    {
        void *ptr = state_mgr_->exportColumn<ArchetypeT, ComponentT>(slot);
        if (export_ptrs_ != nullptr) {
            export_ptrs_[slot] = ptr;
        }
    }

    template <typename SingletonT>
    MADRONA_HOST_API void exportSingleton(int32_t slot)
    {
        void *ptr = state_mgr_->exportSingleton<SingletonT>(slot);
        if (export_ptrs_ != nullptr) {
            export_ptrs_[slot] = ptr;
        }
    }

    template <typename ArchetypeT, typename ComponentT, EnumType EnumT>
    MADRONA_HOST_API void exportColumn(EnumT slot)
    {
        exportColumn<ArchetypeT, ComponentT>((int32_t)slot);
    }

    template <typename SingletonT, EnumType EnumT>
    MADRONA_HOST_API void exportSingleton(EnumT slot)
    {
        exportSingleton<SingletonT>((int32_t)slot);
    }

    // the ray caster configuration of the executor (reference: registerTypes
    // reads it from GPUImplConsts on the device; here registration is host code)
    MADRONA_HOST_API uint32_t raycastOutputResolution() const
    {
        return state_mgr_->renderConfig(0);
    }

    MADRONA_HOST_API bool raycastRGBD() const
    {
        return state_mgr_->renderConfig(1) != 0;
    }

    // most views per world the render configuration promised (0: unknown)
    MADRONA_HOST_API uint32_t raycastMaxViewsPerWorld() const
    {
        return state_mgr_->renderConfig(2);
    }

    // tells the executor's ray caster where its inputs and outputs live (the
    // reference hands pointers to its BVH kernels in BVHParams,
    // src/mw/cuda_exec.cpp)
    template <typename RenderableT, typename CameraT, typename LightT,
              typename OutputT, typename InstanceC, typename MortonC,
              typename LeafC, typename CameraC, typename LightC, typename RGBC,
              typename DepthC>
    MADRONA_HOST_API void setRenderLayout()
    {
        const uint32_t archetypes[4] = {
            TypeTracker::typeID<RenderableT>(), TypeTracker::typeID<CameraT>(),
            TypeTracker::typeID<LightT>(), TypeTracker::typeID<OutputT>(),
        };
        const uint32_t components[7] = {
            TypeTracker::typeID<InstanceC>(), TypeTracker::typeID<MortonC>(),
            TypeTracker::typeID<LeafC>(), TypeTracker::typeID<CameraC>(),
            TypeTracker::typeID<LightC>(), TypeTracker::typeID<RGBC>(),
            TypeTracker::typeID<DepthC>(),
        };
        state_mgr_->setRenderLayout(archetypes, components);
    }

private:
    StateManager *state_mgr_;
    void **export_ptrs_;
};

}
