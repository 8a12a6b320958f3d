// bvc_basetype.hpp -- C++ facade with the reference's BaseType interface over the C ABI of bvc.h.
//
// Mirrors /root/reference/src/BaseType.h:59-74 (constructor, SetBase, LRT, the public fields var_qual,
// depth_total, alt_bases, depth, af_lrt) so that per-site callers written against the reference class --
// bt_f at src/BaseVarC.cpp:612-615 and :642-652, WriteVcf at src/BaseType.cpp:141-234 -- compile against it
// unchanged.  One object = one site = ONE device round trip (bvc_lrt_csr_comb); that is the compatibility path.  The
// throughput path is bvc::BaseTypeBatch below: gather a tile of sites, one call, one record per site.
//
// Header-only; link with -lbvc.  Errors surface as std::runtime_error (the reference's own failure mode is
// an uncaught std::exception, src/BaseVarC.cpp passim); the C ABI underneath never throws.
#ifndef BVC_BASETYPE_HPP
#define BVC_BASETYPE_HPP

#include <cstdint>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "bvc.h"

namespace bvc {

typedef std::vector<int8_t> BaseV;              // src/BaseType.h:18
typedef std::map<int32_t, int32_t> DepM;        // robin_hood map in the reference (src/BaseType.h:21); <= 4 keys

class Context {
 public:
    explicit Context(int device = 0) : ctx_(nullptr)
    {
        const int rc = bvc_create(&ctx_, device);
        if (rc != BVC_OK) throw std::runtime_error("bvc_create failed (" + std::to_string(rc) + "): no usable gfx950 device");
    }
    ~Context() { bvc_destroy(ctx_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    bvc_ctx *get() const { return ctx_; }
    void check(int rc) const
    {
        if (rc != BVC_OK) throw std::runtime_error(std::string("libbvc: ") + bvc_last_error(ctx_));
    }

 private:
    bvc_ctx *ctx_;
};

// Per-thread default context, like the per-thread BaseType objects of the reference (src/BaseVarC.cpp:263-266).
inline Context &default_context()
{
    static thread_local Context ctx(0);
    return ctx;
}

class BaseType {
 public:
    BaseType(BaseV base, BaseV qual, int8_t ref, double minaf, Context *ctx = nullptr)
        : var_qual(0), depth_total(0), bases(std::move(base)), quals(std::move(qual)), ref_base(ref), min_af(minaf),
          ctx_(ctx ? ctx : &default_context()), done_(false)
    {
        if (bases.size() != quals.size()) throw std::invalid_argument("BaseType: bases and quals differ in length");
        depth = {{0, 0}, {1, 0}, {2, 0}, {3, 0}};
    }

    void SetBase(const BaseV &v)
    {
        if (v.size() > 4) throw std::invalid_argument("SetBase: at most four bases");
        for (auto b : v) if (b < 0 || b > 3) throw std::invalid_argument("SetBase: base outside 0..3");
        base_comb = v;
    }

    bool LRT()
    {
        if (done_) throw std::logic_error("BaseType::LRT is single-shot (the reference consumes its vectors, src/BaseType.cpp:76)");
        done_ = true;
        // one device round trip: the site as a one-row ragged batch with its SetBase list (bvc_lrt_csr_comb)
        const int64_t offsets[2] = {0, static_cast<int64_t>(bases.size())};
        static const int8_t none = 0;
        int8_t comb[4] = {0, 0, 0, 0};
        for (size_t i = 0; i < base_comb.size(); ++i) comb[i] = base_comb[i];
        const uint8_t nc = static_cast<uint8_t>(base_comb.size());
        bvc_site_result r;
        ctx_->check(bvc_lrt_csr_comb(ctx_->get(), 1, offsets, bases.empty() ? &none : bases.data(),
                                     quals.empty() ? &none : quals.data(), &ref_base, min_af, comb, &nc, &r, BVC_PTR_HOST));
        record = r;
        var_qual = r.var_qual;
        depth_total = r.depth_total;
        for (int j = 0; j < 4; ++j) depth[j] = r.depth[j];
        for (int i = 0; i < r.n_alt; ++i) {
            alt_bases.push_back(r.alt_base[i]);
            af_lrt.insert({r.alt_base[i], r.af[i]});
        }
        return r.called != 0;
    }

    double var_qual;
    double depth_total;
    BaseV alt_bases;
    DepM depth;
    std::map<int8_t, double> af_lrt;
    bvc_site_result record{};                   // everything the device returned, diagnostics included

 private:
    BaseV bases;
    BaseV quals;
    BaseV base_comb{0, 1, 2, 3};                // src/BaseType.h:79
    const int8_t ref_base;
    const double min_af;
    Context *ctx_;
    bool done_;
};

// min_af exactly as bt_f derives it (src/BaseVarC.cpp:541-543).
inline double caller_min_af(int32_t n_samples_total, double maf)
{
    double m = 100.0 / n_samples_total;
    if (m > 0.001) m = 0.001;
    if (maf < m) m = maf;
    return m;
}

// The batched form: what the successor of bt_s/bt_f calls once per tile of sites.
class BaseTypeBatch {
 public:
    explicit BaseTypeBatch(Context *ctx = nullptr) : ctx_(ctx ? ctx : &default_context()) {}

    // bases/quals: [n_sites][row_stride] host arrays, uncovered samples marked with base = -1.
    std::vector<bvc_site_result> lrt_dense(int64_t n_sites, int64_t n_samples, int64_t row_stride, const int8_t *bases,
                                           const int8_t *quals, const int8_t *ref_base, double min_af) const
    {
        std::vector<bvc_site_result> out(static_cast<size_t>(n_sites));
        ctx_->check(bvc_lrt_dense(ctx_->get(), n_sites, n_samples, row_stride, bases, quals, ref_base, min_af, out.data(),
                                  BVC_PTR_HOST));
        return out;
    }

    // ragged form: exactly the per-site vectors bt_f builds (src/BaseVarC.cpp:550-559), concatenated.
    std::vector<bvc_site_result> lrt_csr(const std::vector<int64_t> &offsets, const int8_t *bases, const int8_t *quals,
                                         const int8_t *ref_base, double min_af) const
    {
        const int64_t n_sites = static_cast<int64_t>(offsets.size()) - 1;
        std::vector<bvc_site_result> out(static_cast<size_t>(n_sites > 0 ? n_sites : 0));
        if (n_sites > 0)
            ctx_->check(bvc_lrt_csr(ctx_->get(), n_sites, offsets.data(), bases, quals, ref_base, min_af, out.data(),
                                    BVC_PTR_HOST));
        return out;
    }

    // --group: per-site overall records plus [n_sites][n_groups] group records (src/BaseVarC.cpp:617-661).
    void lrt_dense_groups(int64_t n_sites, int64_t n_samples, int64_t row_stride, const int8_t *bases, const int8_t *quals,
                          const int8_t *ref_base, double min_af, const uint8_t *group_of_sample, int32_t n_groups,
                          std::vector<bvc_site_result> &out, std::vector<bvc_group_result> &gout) const
    {
        out.resize(static_cast<size_t>(n_sites));
        gout.resize(static_cast<size_t>(n_sites) * static_cast<size_t>(n_groups));
        ctx_->check(bvc_lrt_dense_groups(ctx_->get(), n_sites, n_samples, row_stride, bases, quals, ref_base, min_af,
                                         group_of_sample, n_groups, out.data(), gout.data(), BVC_PTR_HOST));
    }

 private:
    Context *ctx_;
};

}  // namespace bvc

#endif  // BVC_BASETYPE_HPP
