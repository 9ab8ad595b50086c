// tools/io_probe.cpp -- what the file system of the GPU box gives positioned reads and writes (not product code; numbers are
// quoted in DESIGN.md).  io_probe <dir> [MB]: pread / pwrite of one file by 1..16 threads, into and out of ordinary and
// page-locked memory, fresh files against files that already have their pages.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <functional>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

using clk = std::chrono::steady_clock;
static double ms(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); }

template <typename F>
static double med(int reps, F f)
{
    std::vector<double> v;
    for (int r = 0; r < reps; r++)
        v.push_back(f());
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

static void par(int threads, size_t total, size_t sub, const std::function<void(size_t, size_t)>& work)
{
    std::vector<std::thread> t;
    std::atomic<size_t> next{ 0 };
    for (int i = 0; i < threads; i++)
        t.emplace_back([&] {
            for (;;) {
                const size_t at = next.fetch_add(sub);
                if (at >= total)
                    return;
                work(at, std::min(sub, total - at));
            }
        });
    for (auto& x : t)
        x.join();
}

int main(int argc, char** argv)
{
    const std::string dir = argc > 1 ? argv[1] : "/dev/shm";
    const size_t bytes = (size_t)(argc > 2 ? atoi(argv[2]) : 32) << 20;
    const std::string path = dir + "/io_probe.bin", out = dir + "/io_probe.out";
    char* plain = (char*)aligned_alloc(4096, bytes);
    char* pinned = nullptr;
    if (hipHostMalloc((void**)&pinned, bytes, hipHostMallocDefault) != hipSuccess)
        pinned = nullptr;
    memset(plain, 1, bytes);
    if (pinned)
        memset(pinned, 2, bytes);
    {
        int fd = open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (write(fd, plain, bytes) != (ssize_t)bytes)
            return 1;
        close(fd);
    }
    printf("%zu MB on %s\n", bytes >> 20, dir.c_str());
    printf("memcpy plain->plain2: %.2f ms\n", med(5, [&] {
        char* p2 = (char*)aligned_alloc(4096, bytes);
        memset(p2, 0, bytes);
        auto t0 = clk::now();
        memcpy(p2, plain, bytes);
        auto t1 = clk::now();
        free(p2);
        return ms(t0, t1);
    }));
    for (int kind = 0; kind < 2; kind++) {
        char* mem = kind ? pinned : plain;
        if (!mem)
            continue;
        for (int threads : { 1, 2, 4, 8, 16 }) {
            const double rd = med(7, [&] {
                int fd = open(path.c_str(), O_RDONLY);
                auto t0 = clk::now();
                par(threads, bytes, 1 << 20, [&](size_t at, size_t n) { if (pread(fd, mem + at, n, at) != (ssize_t)n) abort(); });
                auto t1 = clk::now();
                close(fd);
                return ms(t0, t1);
            });
            const double wr_fresh = med(7, [&] {
                int fd = open(out.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
                auto t0 = clk::now();
                par(threads, bytes, 1 << 20, [&](size_t at, size_t n) { if (pwrite(fd, mem + at, n, at) != (ssize_t)n) abort(); });
                auto t1 = clk::now();
                close(fd);
                return ms(t0, t1);
            });
            const double wr_over = med(7, [&] {
                int fd = open(out.c_str(), O_WRONLY, 0644);
                auto t0 = clk::now();
                par(threads, bytes, 1 << 20, [&](size_t at, size_t n) { if (pwrite(fd, mem + at, n, at) != (ssize_t)n) abort(); });
                auto t1 = clk::now();
                close(fd);
                return ms(t0, t1);
            });
            const double trunc = med(5, [&] {
                auto t0 = clk::now();
                int fd = open(out.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
                auto t1 = clk::now();
                par(1, bytes, 1 << 22, [&](size_t at, size_t n) { if (pwrite(fd, mem + at, n, at) != (ssize_t)n) abort(); });
                close(fd);
                return ms(t0, t1);
            });
            printf("%s memory, %2d threads: pread %.2f ms (%.1f GB/s) | pwrite fresh file %.2f ms (%.1f GB/s) | pwrite over existing pages %.2f ms (%.1f GB/s) | open(O_TRUNC) of the full file %.2f ms\n",
                kind ? "page-locked" : "ordinary   ", threads, rd, bytes / rd / 1e6, wr_fresh, bytes / wr_fresh / 1e6, wr_over, bytes / wr_over / 1e6, trunc);
            fflush(stdout);
        }
    }
    // ---- a fresh file through a shared mapping: page faults instead of write() (which holds the inode's lock) ----------
    for (int kind = 0; kind < 2; kind++) {
        char* mem = kind ? pinned : plain;
        if (!mem)
            continue;
        for (int threads : { 1, 2, 4, 8, 16 }) {
            for (int populate = 0; populate < 2; populate++) {
                double t_unlink = 0;
                const double wr = med(7, [&] {
                    auto u0 = clk::now();
                    unlink(out.c_str());
                    auto u1 = clk::now();
                    t_unlink = ms(u0, u1);
                    int fd = open(out.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
                    auto t0 = clk::now();
                    if (ftruncate(fd, bytes) != 0)
                        abort();
                    char* map = (char*)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
                    if (map == MAP_FAILED)
                        abort();
#ifdef MADV_POPULATE_WRITE
                    if (populate)
                        (void)madvise(map, bytes, MADV_POPULATE_WRITE);
#endif
                    par(threads, bytes, 1 << 20, [&](size_t at, size_t n) { memcpy(map + at, mem + at, n); });
                    munmap(map, bytes);
                    auto t1 = clk::now();
                    close(fd);
                    return ms(t0, t1);
                });
                printf("%s memory, %2d threads: mmap + memcpy into a fresh file%s %.2f ms (%.1f GB/s); unlink of the full file before it %.2f ms\n",
                    kind ? "page-locked" : "ordinary   ", threads, populate ? " (MADV_POPULATE_WRITE first)" : "", wr, bytes / wr / 1e6, t_unlink);
                fflush(stdout);
            }
        }
    }
    {
        const double wr = med(7, [&] {
            unlink(out.c_str());
            int fd = open(out.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
            auto t0 = clk::now();
            if (fallocate(fd, 0, 0, bytes) != 0)
                abort();
            auto t1 = clk::now();
            if (pwrite(fd, plain, bytes, 0) != (ssize_t)bytes)
                abort();
            auto t2 = clk::now();
            close(fd);
            printf("fallocate %.2f ms, then pwrite %.2f ms\n", ms(t0, t1), ms(t1, t2));
            return ms(t0, t2);
        });
        printf("fallocate + pwrite, 1 thread: %.2f ms\n", wr);
        // several files at once, one thread each: do writes to DIFFERENT files scale?
        for (int files : { 1, 2, 4, 8 }) {
            const size_t each = bytes / 4;
            std::vector<double> v;
            for (int r = 0; r < 5; r++) {
                for (int f = 0; f < files; f++)
                    unlink((out + std::to_string(f)).c_str());
                auto t0 = clk::now();
                std::vector<std::thread> th;
                for (int f = 0; f < files; f++)
                    th.emplace_back([&, f] {
                        int fd = open((out + std::to_string(f)).c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
                        if (pwrite(fd, plain, each, 0) != (ssize_t)each)
                            abort();
                        close(fd);
                    });
                for (auto& x : th)
                    x.join();
                v.push_back(ms(t0, clk::now()));
            }
            std::sort(v.begin(), v.end());
            printf("%d fresh files of %zu MB at once, one writer each: %.2f ms (%.1f GB/s together)\n", files, each >> 20, v[2], files * each / v[2] / 1e6);
            for (int f = 0; f < files; f++)
                unlink((out + std::to_string(f)).c_str());
        }
    }
    // ---- does the kind of page-locked memory matter to the CPU side?  fallocate + pwrite from / pread into each kind -----
    {
        char* pm = nullptr;
        if (hipHostMalloc((void**)&pm, bytes, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess)
            pm = nullptr;
        if (pm)
            memset(pm, 3, bytes);
        const char* names[3] = { "ordinary", "hipHostMallocDefault", "hipHostMallocPortable|Mapped" };
        char* mems[3] = { plain, pinned, pm };
        for (int round = 0; round < 2; round++)
            for (int k = 0; k < 3; k++) {
                if (!mems[k])
                    continue;
                double fa = 0;
                const double wr = med(9, [&] {
                    unlink(out.c_str());
                    int fd = open(out.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
                    auto t0 = clk::now();
                    if (fallocate(fd, 0, 0, bytes) != 0)
                        abort();
                    auto t1 = clk::now();
                    for (size_t at = 0; at < bytes; at += 1 << 20)
                        if (pwrite(fd, mems[k] + at, 1 << 20, at) != (ssize_t)(1 << 20))
                            abort();
                    auto t2 = clk::now();
                    close(fd);
                    fa = ms(t0, t1);
                    return ms(t1, t2);
                });
                const double rd = med(9, [&] {
                    int fd = open(path.c_str(), O_RDONLY);
                    auto t0 = clk::now();
                    par(4, bytes, 1 << 20, [&](size_t at, size_t n) { if (pread(fd, mems[k] + at, n, at) != (ssize_t)n) abort(); });
                    auto t1 = clk::now();
                    close(fd);
                    return ms(t0, t1);
                });
                printf("%-30s pwrite over fallocated pages, 1 thread: %.2f ms (fallocate %.2f) | pread by 4 threads: %.2f ms\n", names[k], wr, fa, rd);
            }
        if (pm)
            (void)hipHostFree(pm);
    }
    // ---- pages first (fallocate), then copies through a shared mapping by several threads (minor faults only) -----------
    for (int populate = 0; populate < 2; populate++)
        for (int threads : { 1, 2, 4, 8, 16 }) {
            double a = 0, b = 0, c = 0, d = 0;
            const double wr = med(7, [&] {
                unlink(out.c_str());
                int fd = open(out.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
                auto t0 = clk::now();
                if (fallocate(fd, 0, 0, bytes) != 0)
                    abort();
                auto t1 = clk::now();
                char* map = (char*)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED | (populate ? MAP_POPULATE : 0), fd, 0);
                if (map == MAP_FAILED)
                    abort();
                auto t2 = clk::now();
                par(threads, bytes, 1 << 20, [&](size_t at, size_t n) { memcpy(map + at, (pinned ? pinned : plain) + at, n); });
                auto t3 = clk::now();
                munmap(map, bytes);
                auto t4 = clk::now();
                close(fd);
                a = ms(t0, t1), b = ms(t1, t2), c = ms(t2, t3), d = ms(t3, t4);
                return ms(t0, t4);
            });
            printf("fallocate + mmap%s + memcpy by %2d threads + munmap: %.2f ms (last: fallocate %.2f, mmap %.2f, memcpy %.2f, munmap %.2f)\n",
                populate ? "(MAP_POPULATE)" : "", threads, wr, a, b, c, d);
            fflush(stdout);
        }
    unlink(path.c_str());
    unlink(out.c_str());
    return 0;
}
