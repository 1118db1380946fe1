// HDF5 input path: the three static functions of the reference's hdfReader
// (src/hdfReader.h:13-15, src/hdfReader.cpp:9-61) behind the C ABI.  libhdf5 is an optional
// run-time dependency: it is dlopen()ed on first use, so the library links and the
// raw-.bin / synthetic paths work on machines without it.
#include <dlfcn.h>

#include <cstdint>
#include <cstdlib>
#include <ios>
#include <mutex>
#include <stdexcept>
#include <string>

#include "../../include/hnsw_amd.h"

namespace hnsw_amd {

namespace {

typedef int64_t hid_t;      // HDF5 >= 1.10
typedef int herr_t;
typedef unsigned long long hsize_t;

struct Hdf5 {
    void* lib = nullptr;
    herr_t (*H5open)();
    hid_t (*H5Fopen)(const char*, unsigned, hid_t);
    herr_t (*H5Fclose)(hid_t);
    hid_t (*H5Dopen2)(hid_t, const char*, hid_t);
    herr_t (*H5Dclose)(hid_t);
    hid_t (*H5Dget_space)(hid_t);
    int (*H5Sget_simple_extent_ndims)(hid_t);
    int (*H5Sget_simple_extent_dims)(hid_t, hsize_t*, hsize_t*);
    herr_t (*H5Sclose)(hid_t);
    herr_t (*H5Dread)(hid_t, hid_t, hid_t, hid_t, hid_t, void*);
    herr_t (*H5Eset_auto2)(hid_t, void*, void*);
    hid_t native_float = -1, native_int = -1;
};

template <class F>
void sym(void* lib, const char* name, F& out)
{
    out = reinterpret_cast<F>(dlsym(lib, name));
    if (!out) throw std::runtime_error(std::string("libhdf5 lacks symbol ") + name);
}

Hdf5& hdf5()
{
    static Hdf5 h;
    static std::once_flag once;
    static std::string err;
    std::call_once(once, [] {
        const char* env = std::getenv("HNSW_AMD_HDF5_LIB");
        const char* names[] = {env, "libhdf5.so", "libhdf5.so.103", "libhdf5_serial.so", "libhdf5_serial.so.103",
                               "/opt/conda/lib/libhdf5.so", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so"};
        for (const char* n : names) {
            if (!n || !*n) continue;
            h.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (h.lib) break;
        }
        if (!h.lib) { err = "HDF5 input needs libhdf5.so (not found; set HNSW_AMD_HDF5_LIB to its path)"; return; }
        try {
            sym(h.lib, "H5open", h.H5open); sym(h.lib, "H5Fopen", h.H5Fopen); sym(h.lib, "H5Fclose", h.H5Fclose);
            sym(h.lib, "H5Dopen2", h.H5Dopen2); sym(h.lib, "H5Dclose", h.H5Dclose); sym(h.lib, "H5Dget_space", h.H5Dget_space);
            sym(h.lib, "H5Sget_simple_extent_ndims", h.H5Sget_simple_extent_ndims);
            sym(h.lib, "H5Sget_simple_extent_dims", h.H5Sget_simple_extent_dims); sym(h.lib, "H5Sclose", h.H5Sclose);
            sym(h.lib, "H5Dread", h.H5Dread); sym(h.lib, "H5Eset_auto2", h.H5Eset_auto2);
            if (h.H5open() < 0) throw std::runtime_error("H5open failed");
            hid_t* f = reinterpret_cast<hid_t*>(dlsym(h.lib, "H5T_NATIVE_FLOAT_g"));
            hid_t* i = reinterpret_cast<hid_t*>(dlsym(h.lib, "H5T_NATIVE_INT_g"));
            if (!f || !i) throw std::runtime_error("libhdf5 lacks the native type ids");
            h.native_float = *f; h.native_int = *i;
            h.H5Eset_auto2(0, nullptr, nullptr);    // errors are reported through return codes here
        } catch (const std::exception& e) { err = e.what(); h.lib = nullptr; }
    });
    if (!h.lib) throw std::ios_base::failure(err);
    return h;
}

struct File {
    Hdf5& h; hid_t f = -1, d = -1;
    File(Hdf5& hh, const char* filename, const char* dataset) : h(hh)
    {
        f = h.H5Fopen(filename, 0u /* H5F_ACC_RDONLY */, 0 /* H5P_DEFAULT */);
        if (f < 0) throw std::ios_base::failure(std::string("cannot open HDF5 file ") + filename);
        d = h.H5Dopen2(f, dataset, 0);
        if (d < 0) { h.H5Fclose(f); f = -1; throw std::ios_base::failure(std::string("no dataset '") + dataset + "' in " + filename); }
    }
    ~File() { if (d >= 0) h.H5Dclose(d); if (f >= 0) h.H5Fclose(f); }
};

}  // namespace

void hdf_get_dimensions(const char* filename, const char* dataset, uint64_t dims[2])
{
    Hdf5& h = hdf5();
    File fl(h, filename, dataset);
    hid_t sp = h.H5Dget_space(fl.d);
    if (sp < 0) throw std::ios_base::failure("H5Dget_space failed");
    int rank = h.H5Sget_simple_extent_ndims(sp);
    hsize_t dd[32] = {0};
    if (rank < 1 || rank > 2 || h.H5Sget_simple_extent_dims(sp, dd, nullptr) < 0) {
        h.H5Sclose(sp);
        throw std::ios_base::failure(std::string("dataset '") + dataset + "' is not a 1- or 2-dimensional array");
    }
    h.H5Sclose(sp);
    dims[0] = dd[0]; dims[1] = rank == 2 ? dd[1] : 1;
}

void hdf_read(const char* filename, const char* dataset, void* data, bool as_float)
{
    Hdf5& h = hdf5();
    File fl(h, filename, dataset);
    // whole dataset, converted by the library to the native type asked for (H5S_ALL = 0)
    if (h.H5Dread(fl.d, as_float ? h.native_float : h.native_int, 0, 0, 0, data) < 0)
        throw std::ios_base::failure(std::string("H5Dread of '") + dataset + "' failed");
}

}  // namespace hnsw_amd
