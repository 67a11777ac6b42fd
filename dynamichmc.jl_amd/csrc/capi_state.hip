// Checkpoint / resume of a context's chain state (include/dhmc.h dhmc_state_bytes / dhmc_export_state / dhmc_import_state).
#include "capi_internal.hpp"

using namespace capi;

extern "C" {

// ---- resume blob: header + raw images of the per-chain arrays -------------------------------
struct BlobHeader {
    uint64_t magic;
    int32_t dim, chains, Dpad, reserved;
};
static const uint64_t BLOB_MAGIC = 0x31434d4844ull;  // "DHMC1"

int dhmc_state_bytes(dhmc_ctx* c, uint64_t* nbytes) {
    if (!c || !nbytes) return DHMC_ERR_INVALID_ARGUMENT;
    const uint64_t C = c->cfg.chains, Dp = c->Dpad;
    *nbytes = sizeof(BlobHeader) + 4 * C * Dp * sizeof(double) + 2 * C * sizeof(double) + C * sizeof(DAState) + 2 * C * sizeof(uint32_t);
    if (c->cfg.metric == DHMC_METRIC_DENSE) *nbytes += (c->per_chain_dense ? C : 1) * 2 * Dp * Dp * sizeof(double);   // dense M⁻¹ and Wᵀ (shared, or one pair per chain)
    return DHMC_OK;
}

static int blob_io(dhmc_ctx* c, char* blob, bool exporting) {
    const size_t C = c->cfg.chains, Dp = c->Dpad;
    char* p = blob + sizeof(BlobHeader);
    auto io = [&](void* dev, size_t bytes) -> hipError_t {
        hipError_t e = exporting ? hipMemcpy(p, dev, bytes, hipMemcpyDeviceToHost) : hipMemcpy(dev, p, bytes, hipMemcpyHostToDevice);
        p += bytes;
        return e;
    };
    HIP_TRY(c, io(c->st.q, C * Dp * sizeof(double)));
    HIP_TRY(c, io(c->st.g, C * Dp * sizeof(double)));
    HIP_TRY(c, io(c->st.minv, C * Dp * sizeof(double)));
    HIP_TRY(c, io(c->st.W, C * Dp * sizeof(double)));
    HIP_TRY(c, io(c->st.lq, C * sizeof(double)));
    HIP_TRY(c, io(c->st.eps, C * sizeof(double)));
    HIP_TRY(c, io(c->st.da, C * sizeof(DAState)));
    HIP_TRY(c, io(c->st.transition, C * sizeof(uint32_t)));
    HIP_TRY(c, io(c->st.status, C * sizeof(uint32_t)));
    if (c->cfg.metric == DHMC_METRIC_DENSE) {
        const size_t nmat = c->per_chain_dense ? C : 1;
        HIP_TRY(c, io(c->d_Minv, nmat * Dp * Dp * sizeof(double)));
        HIP_TRY(c, io(c->d_WT, nmat * Dp * Dp * sizeof(double)));
    }
    return DHMC_OK;
}

int dhmc_export_state(dhmc_ctx* c, void* host_blob, uint64_t nbytes) {
    uint64_t need;
    if (!c || !host_blob || dhmc_state_bytes(c, &need) || nbytes < need) return DHMC_ERR_INVALID_ARGUMENT;
    DHMC_CHECK_USABLE(c);
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    BlobHeader h{BLOB_MAGIC, c->cfg.dim, c->cfg.chains, c->Dpad, DHMC_DETMATH_VERSION};   // `reserved`: the scalar math the chains were advanced with
    std::memcpy(host_blob, &h, sizeof(h));
    return blob_io(c, (char*)host_blob, true);
}

int dhmc_import_state(dhmc_ctx* c, const void* host_blob, uint64_t nbytes) {
    uint64_t need;
    if (!c || !host_blob || dhmc_state_bytes(c, &need) || nbytes < need) return DHMC_ERR_INVALID_ARGUMENT;
    BlobHeader h;
    std::memcpy(&h, host_blob, sizeof(h));
    if (h.magic != BLOB_MAGIC || h.dim != c->cfg.dim || h.chains != c->cfg.chains || h.Dpad != c->Dpad) return DHMC_ERR_INVALID_ARGUMENT;
    // A blob written under another version of the scalar math (include/dhmc_detmath.h) would resume with different bits than the run
    // it continues: refused (0 = a blob from before the field existed, i.e. version 1 or 2 — accepted only when the caller says so).
    if (h.reserved != DHMC_DETMATH_VERSION && !(h.reserved == 0 && std::getenv("DHMC_ACCEPT_UNVERSIONED_STATE"))) {
        c->err = "dhmc_import_state: the blob was written under detmath version " + std::to_string(h.reserved) + ", this library is version " +
                 std::to_string(DHMC_DETMATH_VERSION) + " (0: unversioned; DHMC_ACCEPT_UNVERSIONED_STATE=1 accepts it)";
        return DHMC_ERR_INVALID_ARGUMENT;
    }
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->poisoned = true;    // a partially copied blob is no state
    const int rc = blob_io(c, const_cast<char*>((const char*)host_blob), false);
    if (rc == DHMC_OK) { c->poisoned = false; c->win_n = -1; }   // (a blob carries no metric window)
    return rc;
}

}  // extern "C"
