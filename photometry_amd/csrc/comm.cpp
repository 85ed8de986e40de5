// comm.cpp -- multi-GPU exchange: ONE gather of the per-rank light-curve block over RCCL/xGMI.
//
// The reference's only "distributed backend" is mpi4py point-to-point task messages
// (run_tessphot_mpi.py:74-209); targets are independent, so here they are statically sharded by
// index over the GPUs and the single data-path exchange is the final gather of the output block
// (SURVEY.md section 8e).  xGMI is point-to-point (7 links per GPU), so the gather is issued as
// direct ncclSend/ncclRecv pairs inside one group: the root receives on all 7 inbound links
// concurrently instead of serialising through a ring.
#include "common.h"
#include <rccl/rccl.h>
#include <cstring>

#define TP_NCCL(ctx, call) do { ncclResult_t _r = (call); if (_r != ncclSuccess) { \
	(ctx)->err = std::string(#call) + ": " + ncclGetErrorString(_r); return TP_ERR_COMM; } } while (0)

extern "C" {

int tp_comm_unique_id(char* id_out, int id_len) {
	if (!id_out || id_len < (int)sizeof(ncclUniqueId)) {
		tp_global_err = "tp_comm_unique_id: buffer must hold 128 bytes";
		return TP_ERR_INVALID;
	}
	ncclUniqueId id;
	ncclResult_t r = ncclGetUniqueId(&id);
	if (r != ncclSuccess) {
		tp_global_err = std::string("ncclGetUniqueId: ") + ncclGetErrorString(r);
		return TP_ERR_COMM;
	}
	std::memcpy(id_out, &id, sizeof(id));
	return TP_OK;
}

int tp_comm_init(tp_ctx* ctx, const char* id, int id_len, int rank, int n_ranks) {
	TP_CHECK_CTX(ctx);
	TP_REQUIRE(ctx, id && id_len >= (int)sizeof(ncclUniqueId), "tp_comm_init: bad unique id");
	TP_REQUIRE(ctx, n_ranks >= 1 && rank >= 0 && rank < n_ranks, "tp_comm_init: bad rank / size");
	TP_REQUIRE(ctx, ctx->comm == nullptr, "tp_comm_init: communicator already initialised");
	ncclUniqueId uid;
	std::memcpy(&uid, id, sizeof(uid));
	ncclComm_t comm = nullptr;
	TP_NCCL(ctx, ncclCommInitRank(&comm, n_ranks, uid, rank));
	ctx->comm = comm;
	ctx->comm_rank = rank;
	ctx->comm_size = n_ranks;
	return TP_OK;
}

int tp_comm_destroy(tp_ctx* ctx) {
	if (!ctx || !ctx->comm) return TP_OK;
	(void)hipStreamSynchronize(ctx->stream);
	ncclCommDestroy((ncclComm_t)ctx->comm);
	ctx->comm = nullptr;
	ctx->comm_rank = 0;
	ctx->comm_size = 1;
	return TP_OK;
}

int tp_comm_info(tp_ctx* ctx, int* rank, int* n_ranks) {
	TP_CHECK_CTX(ctx);
	if (rank) *rank = ctx->comm_rank;
	if (n_ranks) *n_ranks = ctx->comm_size;
	return TP_OK;
}

int tp_comm_gather(tp_ctx* ctx, const void* d_send, void* d_recv, uint64_t nbytes_per_rank, int root) {
	TP_CHECK_CTX(ctx);
	TP_REQUIRE(ctx, d_send != nullptr || nbytes_per_rank == 0, "tp_comm_gather: null send buffer");
	const int n = ctx->comm_size, me = ctx->comm_rank;
	TP_REQUIRE(ctx, root >= 0 && root < n, "tp_comm_gather: bad root");
	TP_REQUIRE(ctx, me != root || d_recv != nullptr || nbytes_per_rank == 0, "tp_comm_gather: root needs a receive buffer");
	if (nbytes_per_rank == 0) return TP_OK;
	if (n == 1 || ctx->comm == nullptr) {
		TP_REQUIRE(ctx, n == 1, "tp_comm_gather: communicator not initialised");
		if (d_recv != d_send)
			TP_HIP(ctx, hipMemcpyAsync(d_recv, d_send, (size_t)nbytes_per_rank, hipMemcpyDeviceToDevice, ctx->stream));
		return TP_OK;
	}
	ncclComm_t comm = (ncclComm_t)ctx->comm;
	TP_NCCL(ctx, ncclGroupStart());
	// inside the group no early return: a failed call is remembered, the group is ALWAYS closed, then the first error is
	// reported (a return between ncclGroupStart and ncclGroupEnd would leave the communicator in an open group)
	ncclResult_t first = ncclSuccess;
	const char* what = nullptr;
	if (me == root) {
		for (int r = 0; r < n && first == ncclSuccess; r++) {
			if (r == root) continue;
			ncclResult_t e = ncclRecv(static_cast<char*>(d_recv) + (size_t)r * nbytes_per_rank, (size_t)nbytes_per_rank, ncclChar, r, comm, ctx->stream);
			if (e != ncclSuccess) { first = e; what = "ncclRecv"; }
		}
	} else {
		ncclResult_t e = ncclSend(d_send, (size_t)nbytes_per_rank, ncclChar, root, comm, ctx->stream);
		if (e != ncclSuccess) { first = e; what = "ncclSend"; }
	}
	ncclResult_t e_end = ncclGroupEnd();
	if (first == ncclSuccess && e_end != ncclSuccess) { first = e_end; what = "ncclGroupEnd"; }
	if (first != ncclSuccess) {
		ctx->err = std::string("tp_comm_gather: ") + what + ": " + ncclGetErrorString(first);
		return TP_ERR_COMM;
	}
	if (me == root) {
		char* mine = static_cast<char*>(d_recv) + (size_t)root * nbytes_per_rank;
		if (mine != d_send)
			TP_HIP(ctx, hipMemcpyAsync(mine, d_send, (size_t)nbytes_per_rank, hipMemcpyDeviceToDevice, ctx->stream));
	}
	return TP_OK;
}

int tp_comm_allgather(tp_ctx* ctx, const void* d_send, void* d_recv, uint64_t nbytes_per_rank) {
	TP_CHECK_CTX(ctx);
	if (nbytes_per_rank == 0) return TP_OK;
	TP_REQUIRE(ctx, d_send && d_recv, "tp_comm_allgather: null buffer");
	if (ctx->comm == nullptr) {
		TP_REQUIRE(ctx, ctx->comm_size == 1, "tp_comm_allgather: communicator not initialised");
		if (d_recv != d_send)
			TP_HIP(ctx, hipMemcpyAsync(d_recv, d_send, (size_t)nbytes_per_rank, hipMemcpyDeviceToDevice, ctx->stream));
		return TP_OK;
	}
	TP_NCCL(ctx, ncclAllGather(d_send, d_recv, (size_t)nbytes_per_rank, ncclChar, (ncclComm_t)ctx->comm, ctx->stream));
	return TP_OK;
}

} // extern "C"
