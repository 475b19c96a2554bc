// The host library draws the reference's std::rand() stream in bulk by advancing glibc's generator state in place
// (host/src/device.cpp).  This checker proves it indistinguishable from calling rand(): same outputs, same state
// afterwards, from several seeds and offsets, for lengths around the generator's degree (31) and one alignment's worth.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#include <chrono>
#include <thread>
namespace ndtpso_host { void draw_rand(int32_t* out, size_t n); }
extern "C" void ndtpso_slam_thread_srand(unsigned seed);
extern "C" int ndtpso_slam_thread_rand(void);
int main() {
  // the bulk draw must be indistinguishable from n calls of rand(), from any state
  for (unsigned seed : {1u, 7u, 123456789u}) {
    for (size_t n : {0ul, 1ul, 2ul, 30ul, 31ul, 32ul, 1000ul, 9093ul}) {
      std::srand(seed);
      for (int i = 0; i < 17; ++i) std::rand();
      std::vector<int32_t> want(n), got(n);
      for (auto& v : want) v = std::rand();
      const int next_want = std::rand();
      std::srand(seed);
      for (int i = 0; i < 17; ++i) std::rand();
      ndtpso_host::draw_rand(got.data(), n);
      const int next_got = std::rand();
      if (want != got || next_want != next_got) { std::printf("MISMATCH seed %u n %zu\n", seed, n); return 1; }
    }
  }
  // a thread's private generator (ndtpso_slam_thread_srand) is srand(seed) + rand() on private state: same numbers, and the
  // process-wide generator does not move (run on a thread of its own: the seeding is per thread and for good)
  {
    bool ok = true;
    std::thread th([&ok] {
      for (unsigned seed : {0u, 1u, 42u, 2147483647u, 4000000000u}) {
        std::srand(seed);
        std::vector<int32_t> want(20000);
        for (auto& v : want) v = std::rand();
        std::srand(99u);
        const int global_next = [] { std::srand(99u); return std::rand(); }();
        std::srand(99u);
        ndtpso_slam_thread_srand(seed);
        std::vector<int32_t> got(20000);
        for (size_t i = 0; i < 7; ++i) got[i] = ndtpso_slam_thread_rand();
        ndtpso_host::draw_rand(got.data() + 7, got.size() - 7);
        if (want != got || std::rand() != global_next) {
          std::printf("MISMATCH private generator, seed %u\n", seed);
          ok = false;
        }
      }
    });
    th.join();
    if (!ok) return 1;
  }
  std::vector<int32_t> buf(9093);
  auto t0 = std::chrono::steady_clock::now();
  for (int k = 0; k < 1000; ++k) ndtpso_host::draw_rand(buf.data(), buf.size());
  double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 1000;
  t0 = std::chrono::steady_clock::now();
  for (int k = 0; k < 1000; ++k) for (auto& v : buf) v = std::rand();
  double us2 = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 1000;
  std::printf("ok: bulk %.1f us, rand() loop %.1f us per 9093 draws\n", us, us2);
  return 0;
}
