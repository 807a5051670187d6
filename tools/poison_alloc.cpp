// Debug allocator for torch.cuda.memory.CUDAPluggableAllocator: every allocation is its own hipMalloc filled with 0xFF bytes
// (f32 / bf16 NaN, integer -1), so a kernel that reads memory nobody wrote shows up as NaN in the results or as a wild index.
// Build: hipcc -shared -fPIC -o tools/libpoison_alloc.so tools/poison_alloc.cpp      Use: tools/poison_probe.py
#include <hip/hip_runtime.h>
#include <sys/types.h>
extern "C" void* poison_malloc(ssize_t size, int device, hipStream_t stream) {
  void* p = nullptr;
  if (size <= 0) return nullptr;
  (void)hipSetDevice(device);
  if (hipMalloc(&p, (size_t)size) != hipSuccess) return nullptr;
  (void)hipMemsetAsync(p, 0xFF, (size_t)size, stream);
  return p;
}
extern "C" void poison_free(void* p, ssize_t, int, hipStream_t) {
  if (p) (void)hipFree(p);
}
