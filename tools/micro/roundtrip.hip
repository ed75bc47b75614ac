// roundtrip.hip — developer micro-benchmark: what one device -> host -> device round trip costs on MI355X, i.e. the
// time from the end of a kernel whose result the host needs to the start of the kernel the host launches in reply.
//   copy+sync    kernel; hipMemcpyAsync of 8 bytes to pinned memory; hipStreamSynchronize        (the engine in round 2)
//   mapped+sync  kernel writes the 8 bytes to mapped pinned memory itself; hipStreamSynchronize
//   mapped+spin  the same, the host spins on a sequence word the kernel writes last (no runtime call to wait)
//   event+spin   hipEventRecord after the kernel, the host spins on hipEventQuery
// Each variant runs a chain of `reps` dependent launches; the figure is the wall time per link minus nothing (a link
// is kernel ~2 us + round trip).
//   hipcc --offload-arch=gfx950 -O3 -o roundtrip tools/micro/roundtrip.hip && ./roundtrip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>

__global__ void k_work(unsigned long long* dev, unsigned long long seq) { dev[0] = seq; }

__global__ void k_work_publish(unsigned long long* dev, volatile unsigned long long* host, unsigned long long seq) {
  dev[0] = seq;
  host[1] = seq * 3;  // payload
  __threadfence_system();
  host[0] = seq;  // sequence word, written last
}

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
  const int reps = 2000;
  unsigned long long *dev, *host;
  hipMalloc(&dev, 64);
  hipHostMalloc(&host, 64, hipHostMallocMapped);
  host[0] = host[1] = 0;
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipEvent_t ev;
  hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  for (int i = 0; i < 10; ++i) k_work<<<1, 64, 0, s>>>(dev, 0);
  hipStreamSynchronize(s);

  double t = now_us();
  for (int i = 1; i <= reps; ++i) {
    k_work<<<1, 64, 0, s>>>(dev, i);
    hipMemcpyAsync(host, dev, 8, hipMemcpyDeviceToHost, s);
    hipStreamSynchronize(s);
    if (host[0] != (unsigned long long)i) return 2;
  }
  printf("copy+sync    %7.2f us per link\n", (now_us() - t) / reps);

  t = now_us();
  for (int i = 1; i <= reps; ++i) {
    k_work_publish<<<1, 64, 0, s>>>(dev, host, reps + i);
    hipStreamSynchronize(s);
    if (host[0] != (unsigned long long)(reps + i)) return 3;
  }
  printf("mapped+sync  %7.2f us per link\n", (now_us() - t) / reps);

  t = now_us();
  for (int i = 1; i <= reps; ++i) {
    const unsigned long long seq = 2ull * reps + i;
    k_work_publish<<<1, 64, 0, s>>>(dev, host, seq);
    while (*(volatile unsigned long long*)&host[0] != seq) {
    }
    if (*(volatile unsigned long long*)&host[1] != seq * 3) return 4;
  }
  printf("mapped+spin  %7.2f us per link\n", (now_us() - t) / reps);
  hipStreamSynchronize(s);

  t = now_us();
  for (int i = 1; i <= reps; ++i) {
    k_work<<<1, 64, 0, s>>>(dev, i);
    hipEventRecord(ev, s);
    while (hipEventQuery(ev) == hipErrorNotReady) {
    }
  }
  printf("event+spin   %7.2f us per link\n", (now_us() - t) / reps);

  // a chain of dependent launches without the host in between, for scale
  t = now_us();
  for (int i = 1; i <= reps; ++i) k_work<<<1, 64, 0, s>>>(dev, i);
  hipStreamSynchronize(s);
  printf("no host      %7.2f us per link\n", (now_us() - t) / reps);
  return 0;
}
