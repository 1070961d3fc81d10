// psf_sdma.hpp -- device -> pinned-host copies on the SDMA engines, issued through the HSA runtime that libamdhip64 already has loaded.
//
// Why not hipMemcpyAsync: on this stack the HIP runtime moves a device -> pinned-host copy with a SHADER kernel (__amd_rocclr_copyBuffer in the rocprofv3 kernel
// trace), and a shader copy that stores over PCIe while the next call's FP64 product runs costs that product about what the copy lasts (profiles/r04_notes.md:
// k_trmm_stream 13 -> 19-23 ms, k_trmm_f64_big 39 -> 41-43 ms under the copies of the call before; the same with a copy kernel of our own, any grid size, any queue
// priority).  hsa_amd_memory_async_copy hands the transfer to a DMA engine, which reads HBM and feeds the link without a wave on any CU.
//
// The HSA entry points are looked up in the library HIP has already mapped (dlopen of the same soname returns that instance; hsa_init only adds a reference), so the
// product library's link dependencies stay what they were (libamdhip64 only).  The GPU agent is the one with the HIP device's PCI bus / device number.
#pragma once
#include <dlfcn.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <chrono>
#include <vector>

namespace psf {

struct SdmaCopy {
  void* lib = nullptr;
  bool ready = false;
  hsa_agent_t gpu{}, cpu{};
  decltype(&hsa_init) p_init = nullptr;
  decltype(&hsa_shut_down) p_shut_down = nullptr;
  decltype(&hsa_iterate_agents) p_iterate_agents = nullptr;
  decltype(&hsa_agent_get_info) p_agent_get_info = nullptr;
  decltype(&hsa_signal_create) p_signal_create = nullptr;
  decltype(&hsa_signal_destroy) p_signal_destroy = nullptr;
  decltype(&hsa_signal_store_screlease) p_signal_store = nullptr;
  decltype(&hsa_signal_wait_scacquire) p_signal_wait = nullptr;
  decltype(&hsa_amd_memory_async_copy) p_async_copy = nullptr;

  struct Agents { SdmaCopy* self; std::vector<hsa_agent_t> gpus, cpus; };
  static hsa_status_t collect(hsa_agent_t a, void* data) {
    auto* ag = static_cast<Agents*>(data);
    hsa_device_type_t t;
    if (ag->self->p_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
    if (t == HSA_DEVICE_TYPE_GPU) ag->gpus.push_back(a); else if (t == HSA_DEVICE_TYPE_CPU) ag->cpus.push_back(a);
    return HSA_STATUS_SUCCESS;
  }

  // false (and nothing held) when the HSA runtime or the device's agent cannot be found: the caller then stays on the HIP copies
  bool open(int pci_domain, int pci_bus, int pci_device) {
    if (ready) return true;
    lib = dlopen("libhsa-runtime64.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) return false;
    auto sym = [&](const char* name) { return dlsym(lib, name); };
    p_init = reinterpret_cast<decltype(p_init)>(sym("hsa_init"));
    p_shut_down = reinterpret_cast<decltype(p_shut_down)>(sym("hsa_shut_down"));
    p_iterate_agents = reinterpret_cast<decltype(p_iterate_agents)>(sym("hsa_iterate_agents"));
    p_agent_get_info = reinterpret_cast<decltype(p_agent_get_info)>(sym("hsa_agent_get_info"));
    p_signal_create = reinterpret_cast<decltype(p_signal_create)>(sym("hsa_signal_create"));
    p_signal_destroy = reinterpret_cast<decltype(p_signal_destroy)>(sym("hsa_signal_destroy"));
    p_signal_store = reinterpret_cast<decltype(p_signal_store)>(sym("hsa_signal_store_screlease"));
    p_signal_wait = reinterpret_cast<decltype(p_signal_wait)>(sym("hsa_signal_wait_scacquire"));
    p_async_copy = reinterpret_cast<decltype(p_async_copy)>(sym("hsa_amd_memory_async_copy"));
    if (!p_init || !p_shut_down || !p_iterate_agents || !p_agent_get_info || !p_signal_create || !p_signal_destroy || !p_signal_store || !p_signal_wait || !p_async_copy ||
        p_init() != HSA_STATUS_SUCCESS) { dlclose(lib); lib = nullptr; return false; }
    Agents ag{this, {}, {}};
    bool found = false;
    if (p_iterate_agents(&SdmaCopy::collect, &ag) == HSA_STATUS_SUCCESS && !ag.cpus.empty()) {
      for (hsa_agent_t a : ag.gpus) {
        uint32_t bdf = 0, dom = 0;
        if (p_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf) != HSA_STATUS_SUCCESS) continue;
        p_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &dom);
        if ((int)((bdf >> 8) & 0xff) == pci_bus && (int)((bdf >> 3) & 0x1f) == pci_device && (int)dom == pci_domain) { gpu = a; found = true; break; }
      }
      cpu = ag.cpus[0];
    }
    if (!found) { p_shut_down(); dlclose(lib); lib = nullptr; return false; }
    ready = true;
    return true;
  }

  void close() {
    if (!ready) return;
    p_shut_down();
    dlclose(lib);
    lib = nullptr; ready = false;
  }

  bool make_signal(hsa_signal_t* s) const { return p_signal_create(1, 0, nullptr, s) == HSA_STATUS_SUCCESS; }
  void drop_signal(hsa_signal_t s) const { if (s.handle) p_signal_destroy(s); }

  // device -> pinned host; `done` drops below 1 when the bytes have landed
  bool start(void* host_dst, const void* dev_src, size_t bytes, hsa_signal_t done) const {
    p_signal_store(done, 1);
    return p_async_copy(host_dst, cpu, dev_src, gpu, bytes, 0, nullptr, done) == HSA_STATUS_SUCCESS;
  }

  // pinned host -> device (the targets of the next call, moved while the call before still computes)
  bool start_upload(void* dev_dst, const void* host_src, size_t bytes, hsa_signal_t done) const {
    p_signal_store(done, 1);
    return p_async_copy(dev_dst, gpu, host_src, cpu, bytes, 0, nullptr, done) == HSA_STATUS_SUCCESS;
  }

  // false on a copy error or when nothing arrived within `seconds` (a lost engine must not hang the caller's thread for ever)
  bool wait(hsa_signal_t done, double seconds = 30.0) const {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      const hsa_signal_value_t v = p_signal_wait(done, HSA_SIGNAL_CONDITION_LT, 1, 10000000ull, HSA_WAIT_STATE_BLOCKED);
      if (v < 0) return false;
      if (v < 1) return true;
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > seconds) return false;
    }
  }
};

}  // namespace psf
