// tmpfs_write.cpp -- how fast can 16 host threads fill an output file in /dev/shm?  (would a device-side rendering of the
// survivors, which leaves the host only pwrite()s of finished buffers, lift the end-to-end rate?)  g++ -O2 -pthread; usage: tmpfs_write <GB> <threads>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
// 16 threads fill a tmpfs file of `gb` GB: (a) stores into a MAP_SHARED mapping, (b) pwrite from a private 4 MB buffer
int main(int argc, char **argv)
{
    const size_t gb = argc > 1 ? atoi(argv[1]) : 2, nt = argc > 2 ? atoi(argv[2]) : 8;
    const size_t total = gb << 30, chunk = 4u << 20;
    std::vector<char> src(chunk, 'A');
    for (int mode = 0; mode < 3; ++mode) {
        const char *path = "/dev/shm/wb_test.bin";
        unlink(path);
        int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0600);
        if (ftruncate(fd, total) != 0) return 1;
        char *m = mode == 0 ? (char *)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0) : nullptr;
        auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (size_t t = 0; t < nt; ++t)
            th.emplace_back([&, t] {
                std::vector<char> buf(chunk);
                for (size_t o = t * chunk; o < total; o += nt * chunk) {
                    if (mode == 0) memcpy(m + o, src.data(), chunk);
                    else if (mode == 1) { memcpy(buf.data(), src.data(), chunk); if (pwrite(fd, buf.data(), chunk, o) != (ssize_t)chunk) abort(); }
                    else { if (pwrite(fd, src.data(), chunk, o) != (ssize_t)chunk) abort(); }
                }
            });
        for (auto &x : th) x.join();
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("%s: %zu GB by %zu threads in %.3f s = %.2f GB/s\n", mode == 0 ? "mmap stores" : mode == 1 ? "render to a private buffer + pwrite" : "pwrite only", gb, nt, s, gb / s);
        if (m) munmap(m, total);
        close(fd);
        unlink(path);
    }
    return 0;
}
