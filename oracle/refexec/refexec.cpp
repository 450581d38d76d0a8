// refexec — runs functions of the reference's OWN prebuilt binary (/root/reference/bin/arch_x64/sift) in this process.
//
// TEST INFRASTRUCTURE ONLY (like everything under oracle/): it pins the CPU oracle against the real reference.
// The reference cannot be rebuilt here (Vigra, OpenCV and Boost are absent) and its executable cannot be started
// (the same libraries are DT_NEEDED), but the hot path inside it — Sift::calculate and the sift::alg functions, with
// every Vigra template they use compiled in — only needs libc, libm and libstdc++.  This loader maps the
// executable's two PT_LOAD segments at their link addresses, applies its dynamic relocations against the libraries
// already in this process (symbols of the absent libraries, which only main() uses, are pointed at a trap),
// registers its unwind tables, and calls the functions by their symbol-table addresses with argument objects laid
// out like the reference's classes (sift.hpp:17-40, interestpoint.hpp:13-20, matrix.hpp:14-18, octaveelem.hpp:9-13,
// vigra::MultiArray<2,float> = shape[2], stride[2], pointer, allocator).  Nothing of the reference is copied: the
// binary is read where it lies, outputs go to files given on the command line.
//
//   refexec <reference binary> calculate <in.f32> <w> <h> <dogs> <octaves> <sigma> <k> <subpixel> <out prefix>
//   refexec <reference binary> dogs      <in.f32> <w> <h> <dogs> <octaves> <sigma> <k> <subpixel> <out prefix>
//   refexec <reference binary> blur      <in.f32> <w> <h> <sigma> <out.f32>
//   refexec <reference binary> reduce|increase <in.f32> <w> <h> <sigma> <out prefix>
//   refexec <reference binary> dog       <a.f32> <b.f32> <w> <h> <out.f32>
//   refexec <reference binary> parabola  <lx> <ly> <px> <py> <rx> <ry>
#include <dlfcn.h>
#include <elf.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <map>
#include <new>
#include <string>
#include <vector>

extern "C" void __register_frame(void*);

namespace {

[[noreturn]] void die(const char* what) {
    std::fprintf(stderr, "refexec: %s\n", what);
    std::exit(3);
}

extern "C" void refexec_unresolved() {
    std::fprintf(stderr, "refexec: the reference called a symbol of a library that is absent here\n");
    std::_Exit(4);
}

struct Image {   // vigra::MultiArray<2, float>
    long shape[2];
    long stride[2];
    float* ptr;
    long alloc;
};
static_assert(sizeof(Image) == 48, "MultiArray<2,float> layout");

struct Matrix {  // sift::Matrix<T>: u16 width, u16 height, std::shared_ptr<T>
    uint16_t w, h;
    uint32_t pad;
    void* data;
    void* ctrl;
};
static_assert(sizeof(Matrix) == 24, "Matrix layout");

struct SiftObj {  // sift::Sift
    bool subpixel;
    char pad[3];
    float sigma, k;
    uint16_t dogs, octaves;
    Matrix gaussians, magnitudes, orientations;
};
static_assert(sizeof(SiftObj) == 88, "Sift layout");

struct OctaveElem {
    float scale;
    uint32_t pad;
    Image img;
};
static_assert(sizeof(OctaveElem) == 56, "OctaveElem layout");

struct InterestPoint {
    float scale;
    uint16_t octave, index;
    bool filtered;
    char pad0;
    uint16_t x, y;
    char pad1[2];
    float orientation;
    char pad2[4];
    float *d_begin, *d_end, *d_cap;
};
static_assert(sizeof(InterestPoint) == 48, "InterestPoint layout");

struct PointVec {
    InterestPoint *begin, *end, *cap;
};

struct PointUF {   // sift::Point<u16_t, f32_t>
    uint16_t x;
    char pad[2];
    float y;
};

std::map<std::string, uint64_t> g_symbols;

void load(const char* path) {
    const int fd = open(path, O_RDONLY);
    if (fd < 0) die("cannot open the reference binary");
    struct stat st;
    fstat(fd, &st);
    std::vector<unsigned char> file((size_t)st.st_size);
    if (pread(fd, file.data(), file.size(), 0) != (ssize_t)file.size()) die("short read");
    close(fd);
    const auto* eh = reinterpret_cast<const Elf64_Ehdr*>(file.data());
    if (std::memcmp(eh->e_ident, ELFMAG, SELFMAG) != 0 || eh->e_type != ET_EXEC || eh->e_machine != EM_X86_64) die("not an x86-64 ET_EXEC");
    const auto* ph = reinterpret_cast<const Elf64_Phdr*>(file.data() + eh->e_phoff);
    const Elf64_Dyn* dyn = nullptr;
    for (int i = 0; i < eh->e_phnum; ++i) {
        if (ph[i].p_type == PT_LOAD) {
            const uint64_t lo = ph[i].p_vaddr & ~0xfffull, hi = (ph[i].p_vaddr + ph[i].p_memsz + 0xfff) & ~0xfffull;
            void* m = mmap(reinterpret_cast<void*>(lo), hi - lo, PROT_READ | PROT_WRITE | PROT_EXEC,
                           MAP_PRIVATE | MAP_ANONYMOUS | MAP_FIXED_NOREPLACE, -1, 0);
            if (m != reinterpret_cast<void*>(lo)) die("the reference's link address range is not free in this process");
            std::memcpy(reinterpret_cast<void*>(ph[i].p_vaddr), file.data() + ph[i].p_offset, ph[i].p_filesz);
        } else if (ph[i].p_type == PT_DYNAMIC) {
            dyn = reinterpret_cast<const Elf64_Dyn*>(ph[i].p_vaddr);
        } else if (ph[i].p_type == PT_TLS) {
            die("the reference binary has a TLS segment");
        }
    }
    if (!dyn) die("no PT_DYNAMIC");
    const Elf64_Sym* symtab = nullptr;
    const char* strtab = nullptr;
    const Elf64_Rela *rela = nullptr, *jmprel = nullptr;
    size_t relasz = 0, pltrelsz = 0;
    for (const Elf64_Dyn* d = dyn; d->d_tag != DT_NULL; ++d) {
        switch (d->d_tag) {
            case DT_SYMTAB: symtab = reinterpret_cast<const Elf64_Sym*>(d->d_un.d_ptr); break;
            case DT_STRTAB: strtab = reinterpret_cast<const char*>(d->d_un.d_ptr); break;
            case DT_RELA: rela = reinterpret_cast<const Elf64_Rela*>(d->d_un.d_ptr); break;
            case DT_RELASZ: relasz = d->d_un.d_val; break;
            case DT_JMPREL: jmprel = reinterpret_cast<const Elf64_Rela*>(d->d_un.d_ptr); break;
            case DT_PLTRELSZ: pltrelsz = d->d_un.d_val; break;
            default: break;
        }
    }
    if (!symtab || !strtab) die("no dynamic symbol table");
    auto relocate = [&](const Elf64_Rela* r, size_t bytes) {
        for (size_t i = 0; i < bytes / sizeof(Elf64_Rela); ++i) {
            const uint32_t type = ELF64_R_TYPE(r[i].r_info);
            const Elf64_Sym& s = symtab[ELF64_R_SYM(r[i].r_info)];
            const char* name = strtab + s.st_name;
            auto* where = reinterpret_cast<uint64_t*>(r[i].r_offset);
            // the executable's own definition wins, as it would for the dynamic linker
            void* addr = s.st_shndx != SHN_UNDEF ? reinterpret_cast<void*>(s.st_value) : dlsym(RTLD_DEFAULT, name);
            switch (type) {
                case R_X86_64_JUMP_SLOT:
                case R_X86_64_GLOB_DAT:
                    *where = addr ? reinterpret_cast<uint64_t>(addr)
                                  : (ELF64_ST_BIND(s.st_info) == STB_WEAK ? 0 : reinterpret_cast<uint64_t>(&refexec_unresolved));
                    break;
                case R_X86_64_64:
                    *where = (addr ? reinterpret_cast<uint64_t>(addr) : 0) + (uint64_t)r[i].r_addend;
                    break;
                case R_X86_64_COPY: {
                    void* src = dlsym(RTLD_DEFAULT, name);
                    if (src) std::memcpy(where, src, s.st_size);   // absent library: stays zero, only main() looks
                    break;
                }
                case R_X86_64_RELATIVE: *where = (uint64_t)r[i].r_addend; break;
                default: die("unexpected relocation type");
            }
        }
    };
    if (rela) relocate(rela, relasz);
    if (jmprel) relocate(jmprel, pltrelsz);
    // sections: unwind tables and the full symbol table
    const auto* sh = reinterpret_cast<const Elf64_Shdr*>(file.data() + eh->e_shoff);
    const char* shstr = reinterpret_cast<const char*>(file.data() + sh[eh->e_shstrndx].sh_offset);
    for (int i = 0; i < eh->e_shnum; ++i) {
        if (std::strcmp(shstr + sh[i].sh_name, ".eh_frame") == 0) __register_frame(reinterpret_cast<void*>(sh[i].sh_addr));
        if (sh[i].sh_type == SHT_SYMTAB) {
            const auto* syms = reinterpret_cast<const Elf64_Sym*>(file.data() + sh[i].sh_offset);
            const char* names = reinterpret_cast<const char*>(file.data() + sh[sh[i].sh_link].sh_offset);
            for (size_t k = 0; k < sh[i].sh_size / sizeof(Elf64_Sym); ++k)
                if (ELF64_ST_TYPE(syms[k].st_info) == STT_FUNC && syms[k].st_value) g_symbols[names + syms[k].st_name] = syms[k].st_value;
        }
    }
}

template <class F>
F fn(const char* mangled) {
    auto it = g_symbols.find(mangled);
    if (it == g_symbols.end()) {
        std::fprintf(stderr, "refexec: symbol %s not in the reference binary\n", mangled);
        std::exit(3);
    }
    return reinterpret_cast<F>(it->second);
}

std::vector<float> read_f32(const char* path, size_t n) {
    std::vector<float> v(n);
    FILE* f = std::fopen(path, "rb");
    if (!f || std::fread(v.data(), 4, n, f) != n) die("cannot read input floats");
    std::fclose(f);
    return v;
}

void write_file(const std::string& path, const void* p, size_t bytes) {
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f || std::fwrite(p, 1, bytes, f) != bytes) die("cannot write output");
    std::fclose(f);
}

Image make_image(const std::vector<float>& px, long w, long h) {
    Image im;
    im.shape[0] = w; im.shape[1] = h;
    im.stride[0] = 1; im.stride[1] = w;
    im.ptr = static_cast<float*>(::operator new(px.size() * sizeof(float)));   // the reference may free it (sift.cpp:20-21)
    std::memcpy(im.ptr, px.data(), px.size() * sizeof(float));
    im.alloc = 0;
    return im;
}

std::vector<float> dense(const Image& im) {
    std::vector<float> v((size_t)(im.shape[0] * im.shape[1]));
    for (long y = 0; y < im.shape[1]; ++y)
        for (long x = 0; x < im.shape[0]; ++x) v[(size_t)(x + y * im.shape[0])] = im.ptr[x * im.stride[0] + y * im.stride[1]];
    return v;
}

using BlurFn = void (*)(Image*, const Image*, float);
using DogFn = void (*)(Image*, const Image*, const Image*);
using CalcFn = void (*)(PointVec*, SiftObj*, Image*);
using ParabolaFn = float (*)(const PointUF*, const PointUF*, const PointUF*);

int run(int argc, char** argv) {
    const std::string cmd = argv[2];
    if (cmd == "blur" || cmd == "reduce" || cmd == "increase") {
        if (argc != 8) die("usage");
        const long w = std::atol(argv[4]), h = std::atol(argv[5]);
        const Image in = make_image(read_f32(argv[3], (size_t)(w * h)), w, h);
        Image out{};
        const char* sym = cmd == "blur" ? "_ZN4sift3alg17convolveWithGaussERKN5vigra10MultiArrayILj2EfSaIfEEEf"
                        : cmd == "reduce" ? "_ZN4sift3alg17reduceToNextLevelERKN5vigra10MultiArrayILj2EfSaIfEEEf"
                                          : "_ZN4sift3alg19increaseToNextLevelERKN5vigra10MultiArrayILj2EfSaIfEEEf";
        fn<BlurFn>(sym)(&out, &in, (float)std::atof(argv[6]));
        const std::vector<float> v = dense(out);
        if (cmd == "blur") {
            write_file(argv[7], v.data(), v.size() * 4);
        } else {
            const long dims[2] = {out.shape[0], out.shape[1]};
            write_file(std::string(argv[7]) + ".dims", dims, sizeof(dims));
            write_file(std::string(argv[7]) + ".f32", v.data(), v.size() * 4);
        }
        return 0;
    }
    if (cmd == "dog") {
        if (argc != 8) die("usage");
        const long w = std::atol(argv[5]), h = std::atol(argv[6]);
        const Image a = make_image(read_f32(argv[3], (size_t)(w * h)), w, h), b = make_image(read_f32(argv[4], (size_t)(w * h)), w, h);
        Image out{};
        fn<DogFn>("_ZN4sift3alg3dogERKN5vigra10MultiArrayILj2EfSaIfEEES6_")(&out, &a, &b);
        const std::vector<float> v = dense(out);
        write_file(argv[7], v.data(), v.size() * 4);
        return 0;
    }
    if (cmd == "parabola") {
        if (argc != 9) die("usage");
        PointUF p[3]{};
        for (int i = 0; i < 3; ++i) { p[i].x = (uint16_t)std::atoi(argv[3 + 2 * i]); p[i].y = (float)std::atof(argv[4 + 2 * i]); }
        const float r = fn<ParabolaFn>("_ZN4sift3alg14vertexParabolaERKNS_5PointItfEES4_S4_")(&p[0], &p[1], &p[2]);
        uint32_t bits;
        std::memcpy(&bits, &r, 4);
        std::printf("%08x\n", bits);
        return 0;
    }
    if (cmd == "calculate") {
        if (argc != 12) die("usage");
        const long w = std::atol(argv[4]), h = std::atol(argv[5]);
        Image img = make_image(read_f32(argv[3], (size_t)(w * h)), w, h);
        SiftObj s{};
        s.dogs = (uint16_t)std::atoi(argv[6]);
        s.octaves = (uint16_t)std::atoi(argv[7]);
        s.sigma = (float)std::atof(argv[8]);
        s.k = (float)std::atof(argv[9]);
        s.subpixel = std::atoi(argv[10]) != 0;
        const std::string out = argv[11];
        PointVec pts{};
        fn<CalcFn>("_ZN4sift4Sift9calculateERN5vigra10MultiArrayILj2EfSaIfEEE")(&pts, &s, &img);
        // records: x, y, octave, index, filtered as u16 x 5 (+ pad), scale, orientation as f32, n descriptors u32
        const size_t n = (size_t)(pts.end - pts.begin);
        std::vector<unsigned char> rec(n * 24);
        std::vector<float> desc;
        for (size_t i = 0; i < n; ++i) {
            const InterestPoint& p = pts.begin[i];
            const uint16_t u[6] = {p.x, p.y, p.octave, p.index, (uint16_t)(p.filtered ? 1 : 0), 0};
            const uint32_t nd = (uint32_t)(p.d_end - p.d_begin);
            std::memcpy(&rec[i * 24], u, 12);
            std::memcpy(&rec[i * 24 + 12], &p.scale, 4);
            std::memcpy(&rec[i * 24 + 16], &p.orientation, 4);
            std::memcpy(&rec[i * 24 + 20], &nd, 4);
            desc.insert(desc.end(), p.d_begin, p.d_end);
        }
        write_file(out + ".points", rec.data(), rec.size());
        write_file(out + ".desc", desc.data(), desc.size() * 4);
        // the Gaussian pyramid the object keeps (Matrix<OctaveElem>: element (x, y) at x * height + y)
        const auto* el = static_cast<const OctaveElem*>(s.gaussians.data);
        std::vector<long> meta = {s.gaussians.w, s.gaussians.h};
        std::vector<float> levels;
        for (int i = 0; el && i < s.gaussians.w * s.gaussians.h; ++i) {
            meta.push_back(el[i].img.shape[0]);
            meta.push_back(el[i].img.shape[1]);
            uint32_t sb;
            std::memcpy(&sb, &el[i].scale, 4);
            meta.push_back((long)sb);
            const std::vector<float> v = dense(el[i].img);
            levels.insert(levels.end(), v.begin(), v.end());
        }
        write_file(out + ".levels_meta", meta.data(), meta.size() * sizeof(long));
        write_file(out + ".levels", levels.data(), levels.size() * 4);
        const long dims[2] = {img.shape[0], img.shape[1]};
        write_file(out + ".image_dims", dims, sizeof(dims));
        // the gradient maps the object keeps (Matrix<MultiArray>), in their final state: the descriptor stage has
        // added to them in place (sift.cpp:_createDecriptors)
        auto dump_images = [&](const Matrix& m, const std::string& tag) {
            const auto* im = static_cast<const Image*>(m.data);
            std::vector<long> mm = {m.w, m.h};
            std::vector<float> px;
            for (int i = 0; im && i < m.w * m.h; ++i) {
                mm.push_back(im[i].shape[0]);
                mm.push_back(im[i].shape[1]);
                if (im[i].shape[0] * im[i].shape[1] > 0) {
                    const std::vector<float> v = dense(im[i]);
                    px.insert(px.end(), v.begin(), v.end());
                }
            }
            write_file(out + "." + tag + "_meta", mm.data(), mm.size() * sizeof(long));
            write_file(out + "." + tag, px.data(), px.size() * 4);
        };
        dump_images(s.magnitudes, "mag");
        dump_images(s.orientations, "ori");
        std::printf("%zu\n", n);
        return 0;
    }
    if (cmd == "dogs") {   // Sift::_createDOGs alone: the DoG pyramid (sift.cpp:_createDOGs)
        if (argc != 12) die("usage");
        const long w = std::atol(argv[4]), h = std::atol(argv[5]);
        Image img = make_image(read_f32(argv[3], (size_t)(w * h)), w, h);
        SiftObj s{};
        s.dogs = (uint16_t)std::atoi(argv[6]);
        s.octaves = (uint16_t)std::atoi(argv[7]);
        s.sigma = (float)std::atof(argv[8]);
        s.k = (float)std::atof(argv[9]);
        s.subpixel = std::atoi(argv[10]) != 0;
        const std::string out = argv[11];
        Matrix dogs{};
        using DogsFn = void (*)(Matrix*, SiftObj*, Image*);
        fn<DogsFn>("_ZN4sift4Sift11_createDOGsERN5vigra10MultiArrayILj2EfSaIfEEE")(&dogs, &s, &img);
        const auto* el = static_cast<const OctaveElem*>(dogs.data);
        std::vector<long> meta = {dogs.w, dogs.h};
        std::vector<float> levels;
        for (int i = 0; el && i < dogs.w * dogs.h; ++i) {
            meta.push_back(el[i].img.shape[0]);
            meta.push_back(el[i].img.shape[1]);
            uint32_t sb;
            std::memcpy(&sb, &el[i].scale, 4);
            meta.push_back((long)sb);
            const std::vector<float> v = dense(el[i].img);
            levels.insert(levels.end(), v.begin(), v.end());
        }
        write_file(out + ".dogs_meta", meta.data(), meta.size() * sizeof(long));
        write_file(out + ".dogs", levels.data(), levels.size() * 4);
        return 0;
    }
    die("unknown command");
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 3) die("usage: refexec <reference binary> <command> ...");
    load(argv[1]);
    try {
        return run(argc, argv);
    } catch (const std::exception& e) {   // vigra::PreconditionViolation is a std::exception
        std::printf("EXCEPTION %s\n", e.what());
        return 5;
    }
}
