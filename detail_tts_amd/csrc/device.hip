// Per-device kernel attributes and capability checks (gfx950 / MI355X).
#include <mutex>
#include <set>
#include <utility>

#include "common.h"

namespace dtts {

void lds_optin(const void* kernel, int bytes) {
    static std::mutex mu;
    static std::set<std::pair<std::pair<int, const void*>, int>> sizes;      // ((device, kernel), bytes) already served
    int dev = 0;
    DTTS_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> g(mu);
    const auto key = std::make_pair(std::make_pair(dev, kernel), bytes);
    if (sizes.count(key)) return;
    // a kernel may be launched with several LDS sizes: the attribute is a maximum, keep the largest one asked for so far
    int cur = 0;
    for (const auto& e : sizes)
        if (e.first == key.first) cur = e.second > cur ? e.second : cur;
    if (bytes > cur) DTTS_CHECK_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    sizes.insert(key);
}

bool device_fits(int wgs, int lds_bytes) {
    int dev = 0, cus = 0, lds = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
    if (hipDeviceGetAttribute(&lds, hipDeviceAttributeSharedMemPerBlockOptin, dev) != hipSuccess || lds <= 0) {
        (void)hipGetLastError();
        if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) return false;
    }
    return cus >= wgs && lds >= lds_bytes;
}

}  // namespace dtts
