// one thread stamps the shader clock counter (s_memtime) and the 100 MHz wall clock: two launches bracket a piece of a stream,
// d(clock64) / d(wall_clock64) is the average shader clock in between (DVFS: the part clocks to its power budget).
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/clock_probe.hip -o tools/libclockprobe.so      (tools/step_clock.py)
#include <hip/hip_runtime.h>
__global__ void clock_probe_kernel(unsigned long long *out) { out[0] = clock64(); out[1] = wall_clock64(); }
extern "C" int clock_probe(unsigned long long *out, void *stream) {
    clock_probe_kernel<<<1, 1, 0, (hipStream_t)stream>>>(out);
    return (int)hipGetLastError();
}
