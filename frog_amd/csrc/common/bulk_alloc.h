// bulk_alloc.h -- std::vector for arrays with an entry per half-link (10^8 entries, 0.4 GB each for the benchmark group).
//
// Two things a plain std::vector<T> costs at that size, both measured in bin/frog's 2 s:
//   * resize() value-initialises: one thread writes zeros into -- and faults in -- every page of an array that all host threads
//     are about to fill in parallel.  construct() without arguments default-initialises instead (nothing, for the integer types
//     used here); resize(n, value) and assign() keep their meaning.
//   * 4 KiB pages: 100 000 page faults per array on the way in and as many pages to give back on the way out (unmapping the
//     layout's 0.8 GB: 0.07-0.1 s, holding the address space's lock meanwhile).  Blocks of 4 MiB and more are aligned to 2 MiB and
//     marked MADV_HUGEPAGE -- where transparent huge pages are "madvise" or "always" that is 512 times fewer of both, elsewhere
//     the call changes nothing.
#pragma once

#include <cstddef>
#include <cstdlib>
#include <memory>
#include <new>
#include <utility>
#include <vector>

#include <sys/mman.h>

namespace frog {

template <class T> struct BulkAlloc {
    using value_type = T;
    BulkAlloc() = default;
    template <class U> BulkAlloc(const BulkAlloc<U> &) {}
    template <class U> struct rebind { using other = BulkAlloc<U>; };

    static constexpr std::size_t HUGE = (std::size_t)2 << 20;

    T *allocate(std::size_t n)
    {
        const std::size_t bytes = n * sizeof(T);
        if (bytes >= 2 * HUGE) {
            const std::size_t len = (bytes + HUGE - 1) / HUGE * HUGE;
            void *p = std::aligned_alloc(HUGE, len);
            if (!p) throw std::bad_alloc();
#ifdef MADV_HUGEPAGE
            (void)madvise(p, len, MADV_HUGEPAGE);
#endif
            return static_cast<T *>(p);
        }
        void *p = std::malloc(bytes ? bytes : 1);
        if (!p) throw std::bad_alloc();
        return static_cast<T *>(p);
    }
    void deallocate(T *p, std::size_t) { std::free(p); }

    template <class U, class... A> void construct(U *p, A &&...a)
    {
        if constexpr (sizeof...(A) == 0) ::new ((void *)p) U; else ::new ((void *)p) U(std::forward<A>(a)...);
    }
    template <class U> bool operator==(const BulkAlloc<U> &) const { return true; }
    template <class U> bool operator!=(const BulkAlloc<U> &) const { return false; }
};

template <class T> using Bulk = std::vector<T, BulkAlloc<T>>;

} // namespace frog
