// djb_host_ops.hip -- the C ABI of libdjb_hip.so, part 2: the operator surface.  Host <-> HBM pipelines of large
// DJB_MEM_HOST batches, the tier-2 worklist bookkeeping of the two-tier kernels, eval / evalp / pdf / sample / evalp_is /
// io_to_hd / query / merl_index / per-pair-parameter entry points, beckmann::lrep, the synthetic-workload generators and
// the device self-tests.  Shared internals: djb_host.hpp.
#include "djb_host.hpp"

using namespace djbh;

namespace {

djb_status eval_common(djb_ctx *ctx, const djb_brdf *b, int64_t n, const djb_vec3_view *i,
                       const djb_vec3_view *o, const djb_params *params, const djb_vec3_view *out_fr,
                       float *out_pdf, int mem, int want);

// ------------------------------------------------------------------ large host batches: both PCIe directions in flight
// A DJB_MEM_HOST batch of >= 2 chunks is cut into chunks of DJB_HOST_PIPE_CHUNK units (default: n/8 clamped
// to [2^19, 2^23]; 0 disables).  The calling thread copies chunk c+1 in and enqueues its kernels on the context's stream
// while a helper thread copies the results of chunk c out on a second stream (two HBM slots).  Each
// thread keeps the one-pageable-copy-at-a-time rule of Staged::copy, and the path is only taken when
// no input array shares a host page with an output array (see Staged::copy on why).  Results are those
// of the unchunked call: every unit is independent and the chunk kernels are the same kernels.
long long host_pipe_chunk(long long n)
{
	if (const char *e = getenv("DJB_HOST_PIPE_CHUNK")) { long long c = atoll(e); return c < 0 ? 0 : c; }
	// default: eight chunks for mid-sized batches (the first copy in and the last copy out are not overlapped:
	// time ~ input time x (1 + 1/(2 chunks)); >= 2^19 units each keeps a chunk's copies well above the per-copy
	// overhead), 2^23 units for large ones (tools/host_path_rate.py: 2^22..2^24 are within 3 %)
	long long c = ((n + 7) / 8 + 4095) & ~4095LL;
	if (c < (1LL << 19)) c = 1LL << 19;
	if (c > (1LL << 23)) c = 1LL << 23;
	return c;
}

struct HostSpan { uintptr_t lo, hi; };
HostSpan span_of(const djb_vec3_view *v, long long n)
{
	const float *a = v->x < v->y ? v->x : v->y; a = a < v->z ? a : v->z;
	const float *z = v->x > v->y ? v->x : v->y; z = z > v->z ? z : v->z;
	return HostSpan{ (uintptr_t)a, (uintptr_t)(z + (n - 1) * v->stride + 1) };
}
bool share_page(HostSpan a, HostSpan b)
{
	// base pages: separately allocated large arrays are usually adjacent mappings, so anything coarser than
	// the real page size would see every pair of arrays as sharing one
	static const uintptr_t PG = (uintptr_t)sysconf(_SC_PAGESIZE);
	return (a.lo & ~(PG - 1)) < ((b.hi + PG - 1) & ~(PG - 1)) && (b.lo & ~(PG - 1)) < ((a.hi + PG - 1) & ~(PG - 1));
}

// one per-unit array of a chunked host batch: a vec3 view (in the caller's layout) or a float array
struct PipeArr {
	const djb_vec3_view *v = nullptr; int layout = 0;   // vec3
	float *f = nullptr; int width = 1;                    // `width` contiguous floats per unit (1 = scalar, 5 = params record)
	float *dev[2] = { nullptr, nullptr };                 // the two HBM slots
	long long C = 0;
	static PipeArr vec(const djb_vec3_view *v) { PipeArr a; a.v = v; a.layout = Staged::layout_of(v); return a; }
	static PipeArr arr(const float *f, int width = 1) { PipeArr a; a.f = const_cast<float *>(f); a.width = width; return a; }
	HostSpan span(long long n) const { return v ? span_of(v, n) : HostSpan{ (uintptr_t)f, (uintptr_t)(f + (size_t)width * n) }; }
	size_t floats_per_unit() const { return v ? 3 : (size_t)width; }
	djb_vec3_view view(int s) const   // device view of slot s (vec3 arrays)
	{
		float *d = dev[s];
		return layout == 0 ? djb_vec3_view{ d, d + 1, d + 2, 3 } : djb_vec3_view{ d, d + C, d + 2 * C, 1 };
	}
	// units [lo, lo + m) between the caller's memory and slot s; one pageable copy at a time (Staged::copy)
	hipError_t move(int s, long long lo, long long m, bool to_dev, hipStream_t st) const
	{
		const hipMemcpyKind k = to_dev ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost;
		float *hp[3], *dp[3]; size_t cnt; int parts;
		if (!v) { hp[0] = f + (size_t)width * lo; dp[0] = dev[s]; cnt = (size_t)width * m; parts = 1; }
		else if (layout == 0) { hp[0] = v->x + 3 * lo; dp[0] = dev[s]; cnt = 3 * (size_t)m; parts = 1; }
		else { hp[0] = v->x + lo; hp[1] = v->y + lo; hp[2] = v->z + lo; dp[0] = dev[s]; dp[1] = dev[s] + C; dp[2] = dev[s] + 2 * C; cnt = (size_t)m; parts = 3; }
		for (int c = 0; c < parts; ++c) {
			hipError_t e = to_dev ? hipMemcpyAsync(dp[c], hp[c], sizeof(float) * cnt, k, st) : hipMemcpyAsync(hp[c], dp[c], sizeof(float) * cnt, k, st);
			if (e != hipSuccess) return e;
			if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
		}
		return hipSuccess;
	}
};

// Runs launch(m, slot) -- which enqueues the kernels of one chunk on ctx->stream, reading ins[*].dev[slot] and
// writing outs[*].dev[slot] -- over all chunks.  Returns DJB_OK with *taken = false when the batch does not
// qualify (the caller then uses the plain copy-in / run / copy-out path).
template <class Launch>
djb_status host_pipeline(djb_ctx *ctx, long long n, std::vector<PipeArr> &ins, std::vector<PipeArr> &outs, Launch launch, bool *taken)
{
	*taken = false;
	const long long C = host_pipe_chunk(n);
	if (C <= 0 || n < 2 * C || n <= SMALL_N) return DJB_OK;
	// tests set DJB_HOST_PIPE_REQUIRE to turn "fell back to the plain path" into an error
	auto skip = [](const char *why) -> djb_status {
		if (getenv("DJB_HOST_PIPE_REQUIRE")) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: chunked host path not taken: %s", why);
		return DJB_OK;
	};
	for (auto *set : { &ins, &outs })
		for (const PipeArr &a : *set) {
			if (a.v ? !Staged::valid(a.v) : !a.f) return DJB_OK;          // the plain path reports the error
			if (a.v && a.layout == 2) return skip("exotic stride (packed on the host)");
		}
	for (const PipeArr &a : ins)
		for (const PipeArr &o : outs)
			if (share_page(a.span(n), o.span(n))) return skip("an input shares a host page with an output");
	if (!ctx->d2h_stream) {
		HIP_TRY(hipStreamCreateWithFlags(&ctx->d2h_stream, hipStreamNonBlocking));
		for (hipEvent_t &e : ctx->pipe_ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
	}
	*taken = true;

	// two slots per array in HBM, recycled through the context's pool
	Staged pool(ctx, C, DJB_MEM_HOST);
	pool.small = false;
	for (auto *set : { &ins, &outs })
		for (PipeArr &a : *set) {
			a.C = C;
			for (int s = 0; s < 2; ++s) {
				djb_status st = pool.alloc(sizeof(float) * a.floats_per_unit() * (size_t)C, (void **)&a.dev[s]);
				if (st != DJB_OK) return st;
			}
		}

	const long long nch = (n + C - 1) / C;
	std::mutex mu;
	std::condition_variable cv;
	long long issued = 0, drained = 0;
	bool abort_ = false;
	hipError_t werr = hipSuccess;
	std::thread drain([&]() {
		hipError_t e = hipSetDevice(ctx->device);
		for (long long c = 0; c < nch; ++c) {
			{
				std::unique_lock<std::mutex> lk(mu);
				cv.wait(lk, [&] { return issued > c || abort_; });
				if (abort_) return;
			}
			const int s = (int)(c & 1);
			const long long lo = c * C, m = n - lo < C ? n - lo : C;
			if (e == hipSuccess) e = hipStreamWaitEvent(ctx->d2h_stream, ctx->pipe_ev[s], 0);
			for (const PipeArr &a : outs)
				if (e == hipSuccess) e = a.move(s, lo, m, false, ctx->d2h_stream);
			{
				std::lock_guard<std::mutex> lk(mu);
				if (e != hipSuccess) werr = e;
				drained = c + 1;
			}
			cv.notify_all();
		}
	});
	auto stop = [&](djb_status st) {
		{ std::lock_guard<std::mutex> lk(mu); abort_ = true; }
		cv.notify_all();
		drain.join();
		(void)hipStreamSynchronize(ctx->stream);
		(void)hipStreamSynchronize(ctx->d2h_stream);
		return st;
	};
	for (long long c = 0; c < nch; ++c) {
		const int s = (int)(c & 1);
		const long long lo = c * C, m = n - lo < C ? n - lo : C;
		hipError_t late;
		{   // slot s is free once chunk c-2 has left
			std::unique_lock<std::mutex> lk(mu);
			cv.wait(lk, [&] { return drained >= c - 1; });
			late = werr;
		}
		if (late != hipSuccess)   // the helper keeps waiting for the remaining chunks: release it before joining
			return stop(fail(DJB_ERR_HIP, "djb_error: staging copy failed (%s) while returning a host batch", hipGetErrorString(late)));
		hipError_t e = hipSuccess;
		for (const PipeArr &a : ins)
			if (e == hipSuccess) e = a.move(s, lo, m, true, ctx->stream);
		if (e != hipSuccess) return stop(fail(DJB_ERR_HIP, "djb_error: staging copy failed (%s) in chunk %lld of a host batch", hipGetErrorString(e), c));
		djb_status st = launch(m, s);
		if (st != DJB_OK) return stop(st);
		if ((e = hipEventRecord(ctx->pipe_ev[s], ctx->stream)) != hipSuccess)
			return stop(fail(DJB_ERR_HIP, "djb_error: hipEventRecord: %s", hipGetErrorString(e)));
		{ std::lock_guard<std::mutex> lk(mu); issued = c + 1; }
		cv.notify_all();
	}
	drain.join();
	if (werr != hipSuccess) {
		(void)hipStreamSynchronize(ctx->stream);
		return fail(DJB_ERR_HIP, "djb_error: staging copy failed (%s) while returning a host batch", hipGetErrorString(werr));
	}
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	return DJB_OK;
}

djb_status eval_host_pipelined(djb_ctx *ctx, const djb_brdf *b, long long n, const djb_vec3_view *i,
                               const djb_vec3_view *o, const djb_params *params, const djb_vec3_view *out_fr,
                               float *out_pdf, int want, bool *taken)
{
	*taken = false;
	const bool wfr = (want & 3) != 0, wpdf = (want & 4) != 0;
	if (!i || !o || (wfr && !out_fr)) return DJB_OK;                      // the plain path reports the error
	std::vector<PipeArr> ins{ PipeArr::vec(i), PipeArr::vec(o) }, outs;
	if (wfr) outs.push_back(PipeArr::vec(out_fr));
	if (wpdf) outs.push_back(PipeArr::arr(out_pdf));
	return host_pipeline(ctx, n, ins, outs, [&](long long m, int s) {
		djb_vec3_view vi = ins[0].view(s), vo = ins[1].view(s), vf = wfr ? outs[0].view(s) : djb_vec3_view{ nullptr, nullptr, nullptr, 0 };
		return eval_common(ctx, b, m, &vi, &vo, params, wfr ? &vf : nullptr, wpdf ? outs.back().dev[s] : nullptr, DJB_MEM_DEVICE, want);
	}, taken);
}

// worklist capacity bookkeeping (ctx->call_mu held).  wl_adapt: if the previous large call has finished and its list
// overflowed, grow the share (never blocks: an unfinished call is looked at next time).  wl_note: remember this call.
void wl_adapt(djb_ctx *ctx)
{
	if (!ctx->wl_pending || hipEventQuery(ctx->wl_ev) != hipSuccess) return;
	ctx->wl_pending = false;
	// sharded lists (contract mode): the fullest shard decides; scaled to the whole list
	unsigned long long count = 0, total = 0;
	for (int k = 0; k < ctx->wl_words; ++k) { count = std::max(count, (unsigned long long)ctx->wl_host[k]); total += ctx->wl_host[k]; }
	count *= (unsigned long long)ctx->wl_words;
	ctx->wl_last_share = ctx->wl_last_n > 0 ? (double)total / (double)ctx->wl_last_n : 0.0;
	if (ctx->wl_note_key) { ctx->ct_key = ctx->wl_note_key; ctx->ct_key_share = ctx->wl_last_share; ctx->wl_note_key = 0; }
	if ((size_t)count > ctx->wl_last_cap && ctx->wl_last_n > 0) {
		const double need = 1.25 * (double)count / (double)ctx->wl_last_n;
		ctx->wl_frac = std::min(0.25, std::max(need, 2.0 * ctx->wl_frac));
	}
}
// true iff THIS call's counters are the ones in flight (a small call, or a note still pending from another call, is not recorded)
bool wl_note(djb_ctx *ctx, const unsigned int *count, size_t cap, long long n, int words = 1, int stride = 1)
{
	if (n < (1LL << 20) || ctx->wl_pending) return false;
	if (!ctx->wl_ev && hipEventCreateWithFlags(&ctx->wl_ev, hipEventDisableTiming) != hipSuccess) { ctx->wl_ev = nullptr; return false; }
	if (!ctx->wl_host && hipHostMalloc((void **)&ctx->wl_host, sizeof(unsigned int) * djbk::CONTRACT_SHARDS) != hipSuccess) { ctx->wl_host = nullptr; return false; }
	// `words` counters, `stride` words apart -> packed into the pinned block (one strided 2-D copy)
	if (hipMemcpy2DAsync(ctx->wl_host, sizeof(unsigned int), count, sizeof(unsigned int) * stride, sizeof(unsigned int), (size_t)words,
	                     hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) return false;
	if (hipEventRecord(ctx->wl_ev, ctx->stream) != hipSuccess) return false;
	ctx->wl_last_cap = cap; ctx->wl_last_n = n; ctx->wl_words = words; ctx->wl_pending = true;
	ctx->wl_note_key = 0;            // a key is attached by the contract path only, and only to its own note
	return true;
}

djb_status eval_common(djb_ctx *ctx, const djb_brdf *b, int64_t n, const djb_vec3_view *i,
                       const djb_vec3_view *o, const djb_params *params, const djb_vec3_view *out_fr,
                       float *out_pdf, int mem, int want)
{
	if (!b) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null brdf");
	if (!ctx) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null ctx");
	djb_status st = cpu_pair_check(ctx, b);
	if (st != DJB_OK) return st;
	if (is_cpu(ctx)) return djbcpu::eval(ctx, b, n, i, o, params, out_fr, out_pdf, want);
	if (const djb_brdf *tw = scalar_twin(ctx, b, n, mem)) return djbcpu::eval(djbcpu::twin_ctx(), tw, n, i, o, params, out_fr, out_pdf, want);
	st = check_call(ctx, b, n, mem);
	if (st != DJB_OK) return st;
	std::lock_guard<std::recursive_mutex> call_lock(ctx->call_mu);
	Params p;
	if ((st = device_params(params, &p, b->dev.kind)) != DJB_OK) return st;
	if (mem == DJB_MEM_HOST && n > SMALL_N) {
		bool taken = false;
		st = eval_host_pipelined(ctx, b, n, i, o, params, out_fr, out_pdf, want, &taken);
		if (taken || st != DJB_OK) return st;
	}
	Staged sg(ctx, n, mem);
	View vi, vo, vout{ nullptr, nullptr, nullptr, 0 };
	float *dpdf = nullptr;
	if ((st = sg.in_vec(i, &vi)) != DJB_OK) return st;
	if ((st = sg.in_vec(o, &vo)) != DJB_OK) return st;
	if ((want & 3) && (st = sg.out_vec(out_fr, &vout)) != DJB_OK) return st;
	if ((want & 4) && (st = sg.out_arr(out_pdf, &dpdf)) != DJB_OK) return st;
	// The two-tier kernels write placeholders in tier 1 and re-read their inputs in tier 2 (worklist overflow: the whole
	// batch), so a device-resident caller whose OUTPUT arrays overlap its INPUT arrays (in-place evalp) must not use them:
	// such calls take the one-kernel forms, which read a pair before they write it (index-aligned in-place views are
	// the only supported kind of overlap, as for any elementwise kernel).
	bool aliased = false;
	if (mem == DJB_MEM_DEVICE) {
		auto span = [&](const float *q, long long stride) { return std::make_pair((uintptr_t)q, (uintptr_t)(q + (n > 0 ? (n - 1) * stride + 1 : 0))); };
		auto hit = [&](const float *a, long long sa, const float *bq, long long sb) {
			if (!a || !bq || n <= 0) return false;
			auto x = span(a, sa), y = span(bq, sb);
			return x.first < y.second && y.first < x.second;
		};
		const float *ins[6] = { vi.x, vi.y, vi.z, vo.x, vo.y, vo.z };
		const long long sin_[6] = { vi.stride, vi.stride, vi.stride, vo.stride, vo.stride, vo.stride };
		const float *outs_[4] = { (want & 3) ? vout.x : nullptr, (want & 3) ? vout.y : nullptr, (want & 3) ? vout.z : nullptr, dpdf };
		const long long sout[4] = { vout.stride, vout.stride, vout.stride, 1 };
		for (int a = 0; a < 4 && !aliased; ++a) for (int c = 0; c < 6; ++c) if (hit(outs_[a], sout[a], ins[c], sin_[c])) { aliased = true; break; }
	}
	if (b->dev.kind == DJB_KIND_MERL && (want & 3) && !ctx->merl_exact_only && !aliased) {
		// two-tier exact lookup (both tiers in one kernel: tier 2 is drained from per-wave LDS queues); pair indices travel as uint32, so very
		// large batches are chunked
		const long long CH = 1LL << 31;
		for (long long lo = 0; lo < n; lo += CH) {
			long long m = n - lo < CH ? n - lo : CH;
			auto off = [&](const View &v) { return View{ v.x + lo * v.stride, v.y + lo * v.stride, v.z + lo * v.stride, v.stride }; };
			View oi = off(vi), oo = off(vo), ou = (want & 3) ? off(vout) : vout;
			HIP_TRY(djbk::launch_merl_twotier(ctx->stream, b->dev, m, oi, oo, ou, dpdf ? dpdf + lo : nullptr, want));
		}
		return sg.finish();
	}
	if (b->dev.kind == DJB_KIND_UTIA && (want & 3) && !ctx->utia_exact_only && !aliased) {
		// two-tier (djb_kernels_eval.hip): pair indices travel as uint32, so very large batches are chunked; the
		// worklist (16-byte header + 4 bytes per entry; ~2e-5 of the pairs need it) shares the context's scratch
		const long long CH = 1LL << 31;
		for (long long lo = 0; lo < n; lo += CH) {
			long long m = n - lo < CH ? n - lo : CH;
			size_t cap = (size_t)(m / 256 + 4096);
			if (ctx->test_worklist_cap >= 0) cap = (size_t)ctx->test_worklist_cap;   // DJB_OPT_TEST_WORKLIST_CAP (tests): force the overflow path
			size_t need = 16 + 4 * (cap ? cap : 1);
			if (ctx->scratch_bytes < need) {
				HIP_TRY(hipStreamSynchronize(ctx->stream));
				if (ctx->scratch) (void)hipFree(ctx->scratch);
				ctx->scratch = nullptr; ctx->scratch_bytes = 0;
				HIP_TRY(hipMalloc(&ctx->scratch, need));
				ctx->scratch_bytes = need;
			}
			unsigned int *count = (unsigned int *)ctx->scratch, *list = count + 4;
			auto off = [&](const View &v) { return View{ v.x + lo * v.stride, v.y + lo * v.stride, v.z + lo * v.stride, v.stride }; };
			HIP_TRY(djbk::launch_utia_twotier(ctx->stream, b->dev, m, off(vi), off(vo), off(vout), dpdf ? dpdf + lo : nullptr, want,
			                                  list, (unsigned int)cap, count, ctx->contract_1e5 != 0));
		}
		return sg.finish();
	}
	// contract mode pays while tier 2 is a small share of the batch.  A narrow Beckmann lobe sends most pairs there (the
	// denormal tail of exp(-r^2): 73 % at alpha = 0.05 on the bench distribution) and would cost tier 1 ON TOP of the exact
	// evaluation: if the last large contract call with the same lobe and parameters listed more than 15-20 % of its pairs, the
	// call goes to the bit-exact kernel directly (which satisfies the contract trivially)
	unsigned long long ct_key = 0;
	{ unsigned int w[4]; float f4[4] = { p.ax, p.ay, p.rho, (float)b->dev.kind }; memcpy(w, f4, 16); ct_key = ((unsigned long long)(w[0] ^ (w[2] * 2654435761u)) << 32) | (w[1] ^ (w[3] * 40503u));
	  // model lobes (sgd / abc) take no params: the object is the key (each material has its own wall)
	  if (b->dev.kind == DJB_KIND_SGD || b->dev.kind == DJB_KIND_ABC) ct_key ^= (unsigned long long)(uintptr_t)b * 0x9E3779B97F4A7C15ull;
	  if (!ct_key) ct_key = 1; }
	wl_adapt(ctx);
	// The verdict is re-examined: every CT_REPROBE-th call with the hopeless key runs the contract kernels again (the direction
	// distribution may have changed), and toggling DJB_OPT_CONTRACT_1E5 forgets it (djb_ctx_set_option).  Under the option the
	// returned BITS may therefore depend on the call history; the values stay within the contract either way.
	constexpr unsigned int CT_REPROBE = 16;
	// thresholds: below the worklist's 25 % capacity limit (a list that overflows makes tier 2 redo the whole batch ON TOP of
	// tier 1); sgd's tier 1 is the most expensive of the set (0.9 ms against 5 ms per 1e8), its break-even is lower
	const double ct_give_up = b->dev.kind == DJB_KIND_SGD ? 0.15 : 0.20;
	bool ct_hopeless = ctx->ct_key == ct_key && ctx->ct_key_share > ct_give_up;
	if (ct_hopeless && ctx->contract_1e5 && ++ctx->ct_hopeless_calls % CT_REPROBE == 0) ct_hopeless = false;
	const double *model_host = b->model_host.empty() ? nullptr : b->model_host.data();
	if (ctx->contract_1e5 && !aliased && !ct_hopeless && djbk::contract_supported(b->dev, p, model_host)) {
		auto al16 = [](const void *q) { return ((uintptr_t)q & 15) == 0; };
		const bool dense16 = vi.stride == 1 && vo.stride == 1 && al16(vi.x) && al16(vi.y) && al16(vi.z) && al16(vo.x) && al16(vo.y) && al16(vo.z) &&
		                     (!(want & 3) || (vout.stride == 1 && al16(vout.x) && al16(vout.y) && al16(vout.z))) && (!(want & 4) || al16(dpdf));
		if (dense16) {
			// as for MERL: pair indices travel as uint32; worklist = 512-byte header (CONTRACT_SHARDS counters) + 32-byte records {k, i, o}
			const long long CH = 1LL << 31;
			for (long long lo = 0; lo < n; lo += CH) {
				long long m = n - lo < CH ? n - lo : CH;
				const size_t REC = 32;
				wl_adapt(ctx);
				// a first call has nothing to adapt to, and an overflow costs a full exact pass ON TOP of tier 1 (sgd: 8 ms instead of 1.4 per 1e8
				// on each of the first two calls -- tools/exp/r04/contract_fixup_share.sh): the kinds whose tier 2 is a few per cent by
				// nature (sgd's wall, Beckmann's exp(-r^2) tail) start at 12 % (3.8 B of scratch per pair) instead of 2 %
				const double floor_frac = (b->dev.kind == DJB_KIND_SGD || b->dev.kind == DJB_KIND_BECKMANN) ? 0.12 : 0.0;
				size_t cap = (size_t)((double)m * std::max(ctx->wl_frac, floor_frac)) + 4096;
				cap = (cap + djbk::CONTRACT_SHARDS - 1) / djbk::CONTRACT_SHARDS * djbk::CONTRACT_SHARDS;     // whole segments
				if (ctx->test_worklist_cap >= 0) cap = ((size_t)ctx->test_worklist_cap / djbk::CONTRACT_SHARDS + 1) * djbk::CONTRACT_SHARDS;
				if (cap > 0xfffffff0ull) cap = 0xfffffff0ull / djbk::CONTRACT_SHARDS * djbk::CONTRACT_SHARDS;
				const size_t HDR = sizeof(unsigned int) * djbk::CONTRACT_SHARDS * djbk::CONTRACT_COUNTER_STRIDE;     // 8 KB of counters
				size_t need = HDR + REC * cap;
				if (ctx->scratch_bytes < need) {
					HIP_TRY(hipStreamSynchronize(ctx->stream));
					if (ctx->scratch) (void)hipFree(ctx->scratch);
					ctx->scratch = nullptr; ctx->scratch_bytes = 0;
					HIP_TRY(hipMalloc(&ctx->scratch, need));
					ctx->scratch_bytes = need;
				}
				unsigned int *count = (unsigned int *)ctx->scratch, *list = count + HDR / sizeof(unsigned int);
				auto off = [&](const View &v) { return View{ v.x ? v.x + lo : nullptr, v.y ? v.y + lo : nullptr, v.z ? v.z + lo : nullptr, v.stride }; };
				HIP_TRY(djbk::launch_eval_contract(ctx->stream, b->dev, p, model_host, m, off(vi), off(vo), off(vout), dpdf ? dpdf + lo : nullptr, want,
				                                   list, (unsigned int)cap, count));
				if (wl_note(ctx, count, cap, m, (int)djbk::CONTRACT_SHARDS, (int)djbk::CONTRACT_COUNTER_STRIDE)) ctx->wl_note_key = ct_key;
			}
			return sg.finish();
		}
	}
	HIP_TRY(djbk::launch_eval(ctx->stream, b->dev, p, n, vi, vo, vout, dpdf, want));
	return sg.finish();
}

} // namespace

extern "C" {

// ---------------------------------------------------------------- the operator surface
djb_status djb_eval_batch(djb_ctx *ctx, const djb_brdf *b, int64_t n, const djb_vec3_view *i,
                          const djb_vec3_view *o, const djb_params *params, const djb_vec3_view *out, int mem)
try {
	return eval_common(ctx, b, n, i, o, params, out, nullptr, mem, 1);
}
DJB_ABI_CATCH
djb_status djb_evalp_batch(djb_ctx *ctx, const djb_brdf *b, int64_t n, const djb_vec3_view *i,
                           const djb_vec3_view *o, const djb_params *params, const djb_vec3_view *out, int mem)
try {
	return eval_common(ctx, b, n, i, o, params, out, nullptr, mem, 2);
}
DJB_ABI_CATCH
djb_status djb_pdf_batch(djb_ctx *ctx, const djb_brdf *b, int64_t n, const djb_vec3_view *i,
                         const djb_vec3_view *o, const djb_params *params, float *out_pdf, int mem)
try {
	return eval_common(ctx, b, n, i, o, params, nullptr, out_pdf, mem, 4);
}
DJB_ABI_CATCH
djb_status djb_eval_pdf_batch(djb_ctx *ctx, const djb_brdf *b, int64_t n, const djb_vec3_view *i,
                              const djb_vec3_view *o, const djb_params *params, int want_cos,
                              const djb_vec3_view *out_fr, float *out_pdf, int mem)
try {
	return eval_common(ctx, b, n, i, o, params, out_fr, out_pdf, mem, want_cos ? 6 : 5);
}
DJB_ABI_CATCH

static djb_status sample_common(djb_ctx *ctx, const djb_brdf *b, int64_t n, const float *u1, const float *u2,
                                const djb_vec3_view *o, const djb_params *params, const djb_vec3_view *out_w,
                                const djb_vec3_view *out_i, float *out_pdf, int mem, bool is)
{
	if (!b) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null brdf");
	if (!ctx) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null ctx");
	djb_status st = cpu_pair_check(ctx, b);
	if (st != DJB_OK) return st;
	if (is_cpu(ctx) || scalar_twin(ctx, b, n, mem)) {
		const bool on_cpu = is_cpu(ctx);
		if (is && (!out_w || !out_pdf)) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
		if (!u1 || !u2) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null input array");
		return djbcpu::sample(on_cpu ? ctx : djbcpu::twin_ctx(), on_cpu ? b : b->twin, n, u1, u2, 0, 0, 0, o, params, is ? out_w : nullptr, out_i, out_pdf);
	}
	st = check_call(ctx, b, n, mem);
	if (st != DJB_OK) return st;
	std::lock_guard<std::recursive_mutex> call_lock(ctx->call_mu);
	Params p;
	if ((st = device_params(params, &p, b->dev.kind)) != DJB_OK) return st;
	if (mem == DJB_MEM_HOST && n > SMALL_N && o && out_i && (!is || out_w)) {   // large host batch: chunked, both PCIe directions busy
		bool taken = false;
		std::vector<PipeArr> ins{ PipeArr::arr(u1), PipeArr::arr(u2), PipeArr::vec(o) }, outs{ PipeArr::vec(out_i) };
		if (is) { outs.push_back(PipeArr::vec(out_w)); outs.push_back(PipeArr::arr(out_pdf)); }
		st = host_pipeline(ctx, n, ins, outs, [&](long long m, int s) {
			djb_vec3_view dvo = ins[2].view(s), dvi = outs[0].view(s), dvw = is ? outs[1].view(s) : djb_vec3_view{ nullptr, nullptr, nullptr, 0 };
			return sample_common(ctx, b, m, ins[0].dev[s], ins[1].dev[s], &dvo, params, is ? &dvw : nullptr, &dvi,
			                     is ? outs[2].dev[s] : nullptr, DJB_MEM_DEVICE, is);
		}, &taken);
		if (taken || st != DJB_OK) return st;
	}
	Staged sg(ctx, n, mem);
	View vo, vi, vw; const float *d1, *d2; float *dpdf = nullptr;
	if ((st = sg.in_f(u1, &d1)) != DJB_OK) return st;
	if ((st = sg.in_f(u2, &d2)) != DJB_OK) return st;
	if ((st = sg.in_vec(o, &vo)) != DJB_OK) return st;
	if ((st = sg.out_vec(out_i, &vi)) != DJB_OK) return st;
	if (is) {
		if ((st = sg.out_vec(out_w, &vw)) != DJB_OK) return st;
		if ((st = sg.out_arr(out_pdf, &dpdf)) != DJB_OK) return st;
	}
	// DJB_OPT_CONTRACT_1E5 reaches `sample` of a Beckmann lobe (directions within 1e-5 per component) and the weight / pdf of
	// evalp_is (GGX, Beckmann) -- for the reference's own direction: the pdf moves by 1e-3 for a 1e-5 change of direction
	HIP_TRY(djbk::launch_sample(ctx->stream, b->dev, p, n, d1, d2, 0, 0, 0, vo, vi, is ? &vw : nullptr, dpdf, ctx->contract_1e5));
	return sg.finish();
}

djb_status djb_sample_batch(djb_ctx *ctx, const djb_brdf *b, int64_t n, const float *u1, const float *u2,
                            const djb_vec3_view *o, const djb_params *params, const djb_vec3_view *out_i, int mem)
try {
	return sample_common(ctx, b, n, u1, u2, o, params, nullptr, out_i, nullptr, mem, false);
}
DJB_ABI_CATCH

djb_status djb_evalp_is_batch(djb_ctx *ctx, const djb_brdf *b, int64_t n, const float *u1, const float *u2,
                              const djb_vec3_view *o, const djb_params *params, const djb_vec3_view *out_w,
                              const djb_vec3_view *out_i, float *out_pdf, int mem)
try {
	return sample_common(ctx, b, n, u1, u2, o, params, out_w, out_i, out_pdf, mem, true);
}
DJB_ABI_CATCH

djb_status djb_sample_rng_batch(djb_ctx *ctx, const djb_brdf *b, int64_t n, uint32_t seed_u1, uint32_t seed_u2,
                                uint64_t start, const djb_vec3_view *o, const djb_params *params,
                                const djb_vec3_view *out_i)
try {
	if (is_cpu(ctx)) {
		if (b && !is_cpu(b)) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: brdf belongs to a GPU context");
		return djbcpu::sample(ctx, b, n, nullptr, nullptr, seed_u1, seed_u2, start, o, params, nullptr, out_i, nullptr);
	}
	if (b && is_cpu(b)) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: brdf belongs to a CPU context");
	if (!b) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null brdf");
	djb_status st = check_call(ctx, b, n, DJB_MEM_DEVICE);
	if (st != DJB_OK) return st;
	std::lock_guard<std::recursive_mutex> call_lock(ctx->call_mu);
	Params p;
	if ((st = device_params(params, &p, b->dev.kind)) != DJB_OK) return st;
	if (!Staged::valid(o) || !Staged::valid(out_i)) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null vec3 view");
	View vo{ o->x, o->y, o->z, (long long)o->stride }, vi{ out_i->x, out_i->y, out_i->z, (long long)out_i->stride };
	HIP_TRY(djbk::launch_sample(ctx->stream, b->dev, p, n, nullptr, nullptr, seed_u1, seed_u2, start, vo, vi, nullptr, nullptr, ctx->contract_1e5));
	return DJB_OK;
}
DJB_ABI_CATCH

static djb_status hd_common(djb_ctx *ctx, int64_t n, const djb_vec3_view *a, const djb_vec3_view *b,
                            const djb_vec3_view *c, const djb_vec3_view *d, int mem, bool inverse)
{
	if (!ctx) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null ctx");
	if (is_cpu(ctx)) return djbcpu::io_hd(ctx, n, a, b, c, d, inverse);
	if (mem == DJB_MEM_HOST && n >= 0 && n <= ctx->host_batch_max && !ctx->scalar_on_device) return djbcpu::io_hd(djbcpu::twin_ctx(), n, a, b, c, d, inverse);
	djb_status st = check_call(ctx, nullptr, n, mem);
	if (st != DJB_OK) return st;
	std::lock_guard<std::recursive_mutex> call_lock(ctx->call_mu);
	Staged sg(ctx, n, mem);
	View va, vb, vc, vd;
	if ((st = sg.in_vec(a, &va)) != DJB_OK) return st;
	if ((st = sg.in_vec(b, &vb)) != DJB_OK) return st;
	if ((st = sg.out_vec(c, &vc)) != DJB_OK) return st;
	if ((st = sg.out_vec(d, &vd)) != DJB_OK) return st;
	HIP_TRY(djbk::launch_io_to_hd(ctx->stream, n, va, vb, vc, vd, inverse));
	return sg.finish();
}
djb_status djb_io_to_hd_batch(djb_ctx *ctx, int64_t n, const djb_vec3_view *i, const djb_vec3_view *o,
                              const djb_vec3_view *h, const djb_vec3_view *d, int mem)
try {
	return hd_common(ctx, n, i, o, h, d, mem, false);
}
DJB_ABI_CATCH
djb_status djb_hd_to_io_batch(djb_ctx *ctx, int64_t n, const djb_vec3_view *h, const djb_vec3_view *d,
                              const djb_vec3_view *i, const djb_vec3_view *o, int mem)
try {
	return hd_common(ctx, n, h, d, i, o, mem, true);
}
DJB_ABI_CATCH

djb_status djb_query_batch(djb_ctx *ctx, const djb_brdf *b, int which, int64_t n, const djb_vec3_view *a,
                           const djb_vec3_view *bb, const djb_vec3_view *c, const djb_params *params,
                           const djb_vec3_view *out, int mem)
try {
	if (!b) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null brdf");
	if (!ctx) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null ctx");
	const int bkind = djb_brdf_kind(b);
	const bool aniso = bkind == DJB_KIND_TABULAR_ANISO;
	const bool model = bkind == DJB_KIND_SGD || bkind == DJB_KIND_ABC;
	const bool model_q = which >= DJB_Q_MODEL_NDF && which <= DJB_Q_MODEL_G1;
	if (model) {
		if (!(model_q || which == DJB_Q_FRESNEL) || (which == DJB_Q_MODEL_G1 && bkind != DJB_KIND_SGD))
			return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: sgd / abc answer ndf, gaf, fresnel (and g1 for sgd) only");
	} else if (model_q)
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: DJB_Q_MODEL_* need an sgd or abc brdf");
	const bool user_ndf = bkind == DJB_KIND_USER;          // a user-defined NDF on the host path: the CPU side knows which radial queries it has
	if (bkind > DJB_KIND_TABULAR && !aniso && !model && !user_ndf)
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: queries need a microfacet brdf");
	if ((which >= DJB_Q_QF2_RADIAL && which <= DJB_Q_QF1) && (bkind == DJB_KIND_TABULAR || aniso))
		return fail(DJB_ERR_NOT_IMPLEMENTED, "djb_error: Not Implemented");          // dj_brdf.h:1854, 1859
	if ((which >= DJB_Q_P22_RADIAL && which <= DJB_Q_QF1) && aniso)
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: tabular_anisotropic is not a radial microfacet");
	if ((which >= DJB_Q_ANISO_PDF1 && which <= DJB_Q_ANISO_QF2) && !aniso)
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: pdf1/cdf1/qf1/pdf2/cdf2/qf2 need a tabular_anisotropic");
	djb_status st = cpu_pair_check(ctx, b);
	if (st != DJB_OK) return st;
	if (!Staged::valid(a) || !Staged::valid(out)) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null vec3 view");
	if (is_cpu(ctx)) return n <= 0 ? DJB_OK : djbcpu::query(ctx, b, which, n, a, bb ? bb : a, c ? c : a, params, out);
	if (const djb_brdf *tw = scalar_twin(ctx, b, n, mem)) return n <= 0 ? DJB_OK : djbcpu::query(djbcpu::twin_ctx(), tw, which, n, a, bb ? bb : a, c ? c : a, params, out);
	st = check_call(ctx, b, n, mem);
	if (st != DJB_OK) return st;
	std::lock_guard<std::recursive_mutex> call_lock(ctx->call_mu);
	Params p;
	if ((st = device_params(params, &p, b->dev.kind)) != DJB_OK) return st;
	Staged sg(ctx, n, mem);
	View va, vb, vc, vo;
	if ((st = sg.in_vec(a, &va)) != DJB_OK) return st;
	vb = va; vc = va;
	if (bb && (st = sg.in_vec(bb, &vb)) != DJB_OK) return st;
	if (c && (st = sg.in_vec(c, &vc)) != DJB_OK) return st;
	if ((st = sg.out_vec(out, &vo)) != DJB_OK) return st;
	HIP_TRY(djbk::launch_query(ctx->stream, b->dev, p, which, n, va, vb, vc, vo));
	return sg.finish();
}
DJB_ABI_CATCH

djb_status djb_merl_index_batch(djb_ctx *ctx, int64_t n, const djb_vec3_view *i, const djb_vec3_view *o,
                                int32_t *out_index, int mem)
try {
	if (!ctx) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null ctx");
	if (is_cpu(ctx)) return djbcpu::merl_index(ctx, n, i, o, out_index);
	if (mem == DJB_MEM_HOST && n >= 0 && n <= ctx->host_batch_max && !ctx->scalar_on_device) return djbcpu::merl_index(djbcpu::twin_ctx(), n, i, o, out_index);
	djb_status st = check_call(ctx, nullptr, n, mem);
	if (st != DJB_OK) return st;
	std::lock_guard<std::recursive_mutex> call_lock(ctx->call_mu);
	Staged sg(ctx, n, mem);
	View vi, vo; int32_t *didx;
	if ((st = sg.in_vec(i, &vi)) != DJB_OK) return st;
	if ((st = sg.in_vec(o, &vo)) != DJB_OK) return st;
	if ((st = sg.out_arr(out_index, &didx)) != DJB_OK) return st;
	HIP_TRY(djbk::launch_merl_index(ctx->stream, n, vi, vo, didx));
	return sg.finish();
}
DJB_ABI_CATCH

djb_status djb_merl_bin_keys_batch(djb_ctx *ctx, int64_t n, const djb_vec3_view *i, const djb_vec3_view *o, uint32_t *out_keys, int mem)
try {
	if (!ctx) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null ctx");
	// the host path has no fast tier: its keys are the exact indices (a key equals merl_index wherever tier 1 is certain -- always, here)
	if (is_cpu(ctx)) return djbcpu::merl_index(ctx, n, i, o, reinterpret_cast<int32_t *>(out_keys));
	if (mem == DJB_MEM_HOST && n >= 0 && n <= ctx->host_batch_max && !ctx->scalar_on_device)
		return djbcpu::merl_index(djbcpu::twin_ctx(), n, i, o, reinterpret_cast<int32_t *>(out_keys));
	djb_status st = check_call(ctx, nullptr, n, mem);
	if (st != DJB_OK) return st;
	std::lock_guard<std::recursive_mutex> call_lock(ctx->call_mu);
	Staged sg(ctx, n, mem);
	View vi, vo; int32_t *dkeys;
	if ((st = sg.in_vec(i, &vi)) != DJB_OK) return st;
	if ((st = sg.in_vec(o, &vo)) != DJB_OK) return st;
	if ((st = sg.out_arr(reinterpret_cast<int32_t *>(out_keys), &dkeys)) != DJB_OK) return st;
	HIP_TRY(djbk::launch_merl_keys(ctx->stream, n, vi, vo, reinterpret_cast<uint32_t *>(dkeys)));
	return sg.finish();
}
DJB_ABI_CATCH

// ---------------------------------------------------------------- the reference's file-static helpers (host scalars)
djb_status djb_helper(int which, const float *in, float *out)
try {
	if (!in || !out) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
	return djbcpu::helper(which, in, out);
}
DJB_ABI_CATCH

// ---------------------------------------------------------------- beckmann::lrep (host scalars)
// dj_brdf.h:1959-2051, float arithmetic in the reference's order (this TU is built with
// -ffp-contract=off).  lrep = {E1, E2, E3, E4, E5}.
djb_status djb_lrep_op(int op, const float *a, const float *b, float x, float y, float *out)
try {
	if (!a || !out) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
	float E1 = a[0], E2 = a[1], E3 = a[2], E4 = a[3], E5 = a[4];
	const float dflt[5] = { 0, 0, 1, 1, 0 };
	const float *r = b ? b : dflt;
	switch (op) {
	case DJB_LREP_ADD:                                                  // operator+, :1992-1999
		out[0] = E1 + r[0]; out[1] = E2 + r[1];
		out[2] = E3 + r[2] + 2.0f * E1 * r[0];
		out[3] = E4 + r[3] + 2.0f * E2 * r[1];
		out[4] = E5 + r[4] + E1 * r[1] + E2 * r[0];
		return DJB_OK;
	case DJB_LREP_MUL: case DJB_LREP_IMUL: {                            // operator*, *=, :2001-2033
		if (!(x >= 0.0f)) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: Invalid scale");
		float s2 = x * x;
		out[0] = E1 * x; out[1] = E2 * x; out[2] = E3 * s2; out[3] = E4 * s2; out[4] = E5 * s2;
		return DJB_OK;
	}
	case DJB_LREP_IADD:                                                 // operator+=, :2011-2020 (uses the
		E1 += r[0]; E2 += r[1];                                         //  already-updated E1/E2: kept)
		E3 += r[2] + 2.0f * E1 * r[0];
		E4 += r[3] + 2.0f * E2 * r[1];
		E5 += r[4] + E1 * r[1] + E2 * r[0];
		break;
	case DJB_LREP_SHEAR:                                                // :2035-2042
		E1 += x; E2 += y; E3 += x * x; E4 += y * y; E5 += x * y;
		break;
	case DJB_LREP_SCALE:                                                // :2044-2051
		E1 *= x; E2 *= y; E3 *= x * x; E4 *= y * y; E5 *= x * y;
		break;
	default:
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: unknown lrep op %d", op);
	}
	out[0] = E1; out[1] = E2; out[2] = E3; out[3] = E4; out[4] = E5;
	return DJB_OK;
}
DJB_ABI_CATCH

djb_status djb_params_to_lrep(const djb_params *params, float *out)                      // :1965-1974
try {
	if (!out) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
	djb_params_resolved r;
	djb_status st = resolve_params(params, &r);
	if (st != DJB_OK) return st;
	out[0] = r.tx_n; out[1] = r.ty_n;
	out[2] = 0.5f * r.ax * r.ax + r.tx_n * r.tx_n;
	out[3] = 0.5f * r.ay * r.ay + r.ty_n * r.ty_n;
	out[4] = 0.5f * r.rho * r.ax * r.ay + r.tx_n * r.ty_n;
	return DJB_OK;
}
DJB_ABI_CATCH

djb_status djb_lrep_to_params(const float *l, djb_params *out)                           // :1976-1990
try {
	if (!l || !out) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
	// djb::max(a, b) = a > b ? a : b and djb::min(a, b) = a < b ? a : b with the reference's operand order: a NaN moment stays NaN
	float t1 = l[2] - l[0] * l[0], t2 = l[3] - l[1] * l[1];
	t1 = 0.0f > t1 ? 0.0f : t1; t2 = 0.0f > t2 ? 0.0f : t2;
	double sx = std::sqrt(2.0 * (double)t1), sy = std::sqrt(2.0 * (double)t2);
	float ax = (float)(1e-5 > sx ? 1e-5 : sx), ay = (float)(1e-5 > sy ? 1e-5 : sy);
	float rho = 2.0f * (l[4] - l[0] * l[1]) / (ax * ay);
	rho = -0.99f > rho ? -0.99f : rho; rho = 0.99f < rho ? 0.99f : rho;
	out->kind = DJB_PARAMS_PDFPARAMS;
	out->v[0] = ax; out->v[1] = ay; out->v[2] = rho; out->v[3] = l[0]; out->v[4] = l[1];
	return DJB_OK;
}
DJB_ABI_CATCH

static djb_status eval_pp_common(djb_ctx *ctx, const djb_brdf *b, int64_t n, const djb_vec3_view *i,
                                 const djb_vec3_view *o, const float *rec, int mode, const float *base5,
                                 float scale, int lean_flags, int want, const djb_vec3_view *out_fr, float *out_pdf, float *out_pp, int mem)
{
	if (!b || !rec || !ctx) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
	const int bkind = djb_brdf_kind(b);
	if (bkind > DJB_KIND_TABULAR && bkind != DJB_KIND_TABULAR_ANISO)
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: per-pair params need a microfacet brdf");
	if (want != 1 && want != 2 && want != 4 && want != 5 && want != 6)
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: want must be eval(1)|evalp(2) and/or pdf(4)");
	djb_status st = cpu_pair_check(ctx, b);
	if (st != DJB_OK) return st;
	if (is_cpu(ctx)) return n <= 0 ? DJB_OK : djbcpu::eval_pp(ctx, b, n, i, o, rec, mode, base5, scale, lean_flags, want, out_fr, out_pdf, out_pp);
	if (const djb_brdf *tw = scalar_twin(ctx, b, n, mem)) return n <= 0 ? DJB_OK : djbcpu::eval_pp(djbcpu::twin_ctx(), tw, n, i, o, rec, mode, base5, scale, lean_flags, want, out_fr, out_pdf, out_pp);
	st = check_call(ctx, b, n, mem);
	if (st != DJB_OK) return st;
	std::lock_guard<std::recursive_mutex> call_lock(ctx->call_mu);
	if (mem == DJB_MEM_HOST && n > SMALL_N && i && o && (!(want & 3) || out_fr)) {   // large host batch: chunked, both PCIe directions busy
		bool taken = false;
		const bool wfr = (want & 3) != 0, wpdf = (want & 4) != 0;
		std::vector<PipeArr> ins{ PipeArr::vec(i), PipeArr::vec(o), PipeArr::arr(rec, 5) }, outs;
		int kf = -1, kp = -1, kq = -1;
		if (wfr) { kf = (int)outs.size(); outs.push_back(PipeArr::vec(out_fr)); }
		if (wpdf) { kp = (int)outs.size(); outs.push_back(PipeArr::arr(out_pdf)); }
		if (out_pp) { kq = (int)outs.size(); outs.push_back(PipeArr::arr(out_pp, 5)); }
		st = host_pipeline(ctx, n, ins, outs, [&](long long m, int s) {
			djb_vec3_view dvi = ins[0].view(s), dvo = ins[1].view(s), dvf = wfr ? outs[kf].view(s) : djb_vec3_view{ nullptr, nullptr, nullptr, 0 };
			return eval_pp_common(ctx, b, m, &dvi, &dvo, ins[2].dev[s], mode, base5, scale, lean_flags, want, wfr ? &dvf : nullptr,
			                      wpdf ? outs[kp].dev[s] : nullptr, out_pp ? outs[kq].dev[s] : nullptr, DJB_MEM_DEVICE);
		}, &taken);
		if (taken || st != DJB_OK) return st;
	}
	Staged sg(ctx, n, mem);
	View vi, vo, vout{ nullptr, nullptr, nullptr, 0 };
	float *dpdf = nullptr, *dpp = nullptr; const float *drec = rec;
	if ((st = sg.in_vec(i, &vi)) != DJB_OK) return st;
	if ((st = sg.in_vec(o, &vo)) != DJB_OK) return st;
	if (mem == DJB_MEM_HOST) {
		float *d = nullptr;
		if ((st = sg.alloc(sizeof(float) * 5 * (size_t)n, (void **)&d)) != DJB_OK) return st;
		if (n && (st = sg.copy(d, rec, sizeof(float) * 5 * (size_t)n, hipMemcpyHostToDevice)) != DJB_OK) return st;
		drec = d;
	}
	if ((want & 3) && (st = sg.out_vec(out_fr, &vout)) != DJB_OK) return st;
	if ((want & 4) && (st = sg.out_arr(out_pdf, &dpdf)) != DJB_OK) return st;
	if (out_pp) {
		if (mem == DJB_MEM_DEVICE) dpp = out_pp;
		else {
			if ((st = sg.alloc(sizeof(float) * 5 * (size_t)n, (void **)&dpp)) != DJB_OK) return st;
			sg.out_raw.push_back({ dpp, { out_pp, sizeof(float) * 5 * (size_t)n } });
		}
	}
	HIP_TRY(djbk::launch_eval_pp(ctx->stream, b->dev, n, vi, vo, drec, mode, base5, scale, lean_flags, vout, dpdf, dpp, want));
	return sg.finish();
}

djb_status djb_eval_pp_batch(djb_ctx *ctx, const djb_brdf *b, int64_t n, const djb_vec3_view *i,
                             const djb_vec3_view *o, const float *pdfparams, int want,
                             const djb_vec3_view *out_fr, float *out_pdf, int mem)
try {
	return eval_pp_common(ctx, b, n, i, o, pdfparams, 0, nullptr, 1.0f, 0, want, out_fr, out_pdf, nullptr, mem);
}
DJB_ABI_CATCH

djb_status djb_eval_lean_batch(djb_ctx *ctx, const djb_brdf *b, int64_t n, const djb_vec3_view *i,
                               const djb_vec3_view *o, const djb_params *base, float scale, int lean_flags,
                               const float *lean, int want, const djb_vec3_view *out_fr, float *out_pdf,
                               float *out_pdfparams, int mem)
try {
	if (lean_flags & ~(DJB_LEAN_NAIVE_MIP | DJB_LEAN_BIASED))
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: unknown LEAN flag");
	if (!(scale >= 0.0f)) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: Invalid scale");   // lrep::operator*= asserts sc >= 0, dj_brdf.h:2024
	float base5[5];
	djb_status st = djb_params_to_lrep(base, base5);                                           // lrep2, dj_beckmannconductor.cpp:312
	if (st != DJB_OK) return st;
	return eval_pp_common(ctx, b, n, i, o, lean, 1, base5, scale, lean_flags, want, out_fr, out_pdf, out_pdfparams, mem);
}
DJB_ABI_CATCH

// sample (out_w == NULL) / evalp_is with per-pair parameter records: mode 0 pdfparams, mode 1 LEAN texels
static djb_status sample_pp_common(djb_ctx *ctx, const djb_brdf *b, int64_t n, const float *u1, const float *u2,
                                   const djb_vec3_view *o, const float *rec, int mode, const float *base5, float scale,
                                   int lean_flags, const djb_vec3_view *out_w, const djb_vec3_view *out_i, float *out_pdf,
                                   float *out_pp, int mem)
{
	if (!b || !rec || !ctx || !u1 || !u2) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
	const int bkind = djb_brdf_kind(b);
	if (bkind > DJB_KIND_TABULAR && bkind != DJB_KIND_TABULAR_ANISO)
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: per-pair params need a microfacet brdf");
	const bool is = out_w != nullptr;
	if (is && !out_pdf) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
	djb_status st = cpu_pair_check(ctx, b);
	if (st != DJB_OK) return st;
	if (is_cpu(ctx)) return n <= 0 ? DJB_OK : djbcpu::sample_pp(ctx, b, n, u1, u2, o, rec, mode, base5, scale, lean_flags, out_w, out_i, out_pdf, out_pp);
	if (const djb_brdf *tw = scalar_twin(ctx, b, n, mem))
		return n <= 0 ? DJB_OK : djbcpu::sample_pp(djbcpu::twin_ctx(), tw, n, u1, u2, o, rec, mode, base5, scale, lean_flags, out_w, out_i, out_pdf, out_pp);
	st = check_call(ctx, b, n, mem);
	if (st != DJB_OK) return st;
	std::lock_guard<std::recursive_mutex> call_lock(ctx->call_mu);
	if (mem == DJB_MEM_HOST && n > SMALL_N && o && out_i) {          // large host batch: chunked, both PCIe directions busy
		bool taken = false;
		std::vector<PipeArr> ins{ PipeArr::arr(u1), PipeArr::arr(u2), PipeArr::vec(o), PipeArr::arr(rec, 5) }, outs{ PipeArr::vec(out_i) };
		int kq = -1;
		if (is) { outs.push_back(PipeArr::vec(out_w)); outs.push_back(PipeArr::arr(out_pdf)); }
		if (out_pp) { kq = (int)outs.size(); outs.push_back(PipeArr::arr(out_pp, 5)); }
		st = host_pipeline(ctx, n, ins, outs, [&](long long m, int s) {
			djb_vec3_view dvo = ins[2].view(s), dvi = outs[0].view(s), dvw = is ? outs[1].view(s) : djb_vec3_view{ nullptr, nullptr, nullptr, 0 };
			return sample_pp_common(ctx, b, m, ins[0].dev[s], ins[1].dev[s], &dvo, ins[3].dev[s], mode, base5, scale, lean_flags,
			                        is ? &dvw : nullptr, &dvi, is ? outs[2].dev[s] : nullptr, out_pp ? outs[kq].dev[s] : nullptr, DJB_MEM_DEVICE);
		}, &taken);
		if (taken || st != DJB_OK) return st;
	}
	Staged sg(ctx, n, mem);
	View vo, vi, vw; const float *d1, *d2, *drec = rec; float *dpdf = nullptr, *dpp = nullptr;
	if ((st = sg.in_f(u1, &d1)) != DJB_OK) return st;
	if ((st = sg.in_f(u2, &d2)) != DJB_OK) return st;
	if ((st = sg.in_vec(o, &vo)) != DJB_OK) return st;
	if (mem == DJB_MEM_HOST) {
		float *d = nullptr;
		if ((st = sg.alloc(sizeof(float) * 5 * (size_t)n, (void **)&d)) != DJB_OK) return st;
		if (n && (st = sg.copy(d, rec, sizeof(float) * 5 * (size_t)n, hipMemcpyHostToDevice)) != DJB_OK) return st;
		drec = d;
	}
	if ((st = sg.out_vec(out_i, &vi)) != DJB_OK) return st;
	if (is) {
		if ((st = sg.out_vec(out_w, &vw)) != DJB_OK) return st;
		if ((st = sg.out_arr(out_pdf, &dpdf)) != DJB_OK) return st;
	}
	if (out_pp) {
		if (mem == DJB_MEM_DEVICE) dpp = out_pp;
		else {
			if ((st = sg.alloc(sizeof(float) * 5 * (size_t)n, (void **)&dpp)) != DJB_OK) return st;
			sg.out_raw.push_back({ dpp, { out_pp, sizeof(float) * 5 * (size_t)n } });
		}
	}
	HIP_TRY(djbk::launch_sample_pp(ctx->stream, b->dev, n, d1, d2, vo, drec, mode, base5, scale, lean_flags, vi, is ? &vw : nullptr, dpdf, dpp));
	return sg.finish();
}

djb_status djb_sample_pp_batch(djb_ctx *ctx, const djb_brdf *b, int64_t n, const float *u1, const float *u2,
                               const djb_vec3_view *o, const float *pdfparams, const djb_vec3_view *out_w,
                               const djb_vec3_view *out_i, float *out_pdf, int mem)
try {
	return sample_pp_common(ctx, b, n, u1, u2, o, pdfparams, 0, nullptr, 1.0f, 0, out_w, out_i, out_pdf, nullptr, mem);
}
DJB_ABI_CATCH

djb_status djb_sample_lean_batch(djb_ctx *ctx, const djb_brdf *b, int64_t n, const float *u1, const float *u2,
                                 const djb_vec3_view *o, const djb_params *base, float scale, int lean_flags,
                                 const float *lean, const djb_vec3_view *out_w, const djb_vec3_view *out_i,
                                 float *out_pdf, float *out_pdfparams, int mem)
try {
	if (lean_flags & ~(DJB_LEAN_NAIVE_MIP | DJB_LEAN_BIASED))
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: unknown LEAN flag");
	if (!(scale >= 0.0f)) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: Invalid scale");
	float base5[5];
	djb_status st = djb_params_to_lrep(base, base5);
	if (st != DJB_OK) return st;
	return sample_pp_common(ctx, b, n, u1, u2, o, lean, 1, base5, scale, lean_flags, out_w, out_i, out_pdf, out_pdfparams, mem);
}
DJB_ABI_CATCH

djb_status djb_merl_guard_stats(djb_ctx *ctx, int64_t n, const djb_vec3_view *i, const djb_vec3_view *o,
                                const float *guard6, float *max_ratio3, unsigned long long *counters4)
try {
	if (is_cpu(ctx)) return fail(DJB_ERR_NOT_IMPLEMENTED, "djb_error: this diagnostic needs a GPU context");
	djb_status st = check_call(ctx, nullptr, n, DJB_MEM_DEVICE);
	if (st != DJB_OK) return st;
	std::lock_guard<std::recursive_mutex> call_lock(ctx->call_mu);
	if (!Staged::valid(i) || !Staged::valid(o) || !max_ratio3 || !counters4)
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
	unsigned char *d = nullptr;
	HIP_TRY(hipMalloc((void **)&d, 64));
	hipError_t e = hipMemsetAsync(d, 0, 64, ctx->stream);
	if (e == hipSuccess)
		e = djbk::launch_merl_guard_stats(ctx->stream, n, View{ i->x, i->y, i->z, (long long)i->stride },
		                                  View{ o->x, o->y, o->z, (long long)o->stride }, guard6,
		                                  (unsigned int *)d, (unsigned long long *)(d + 16));
	unsigned char h[64];
	if (e == hipSuccess) e = hipMemcpyAsync(h, d, 64, hipMemcpyDeviceToHost, ctx->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
	(void)hipFree(d);
	if (e != hipSuccess) return fail(DJB_ERR_HIP, "djb_error: guard stats: %s", hipGetErrorString(e));
	memcpy(max_ratio3, h, 12);
	memcpy(counters4, h + 16, 32);
	return DJB_OK;
}
DJB_ABI_CATCH

djb_status djb_merl_guard_attack(djb_ctx *ctx, int64_t n, const djb_vec3_view *i, const djb_vec3_view *o, const float *guard6,
                                 int iters, uint32_t seed, float *best_ratio, unsigned long long *counters3)
try {
	if (is_cpu(ctx)) return fail(DJB_ERR_NOT_IMPLEMENTED, "djb_error: this diagnostic needs a GPU context");
	djb_status st = check_call(ctx, nullptr, n, DJB_MEM_DEVICE);
	if (st != DJB_OK) return st;
	std::lock_guard<std::recursive_mutex> call_lock(ctx->call_mu);
	if (!Staged::valid(i) || !Staged::valid(o) || !best_ratio || !counters3 || iters < 0)
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: invalid argument");
	unsigned long long *d = nullptr;
	HIP_TRY(hipMalloc((void **)&d, 32));
	hipError_t e = hipMemsetAsync(d, 0, 32, ctx->stream);
	if (e == hipSuccess)
		e = djbk::launch_merl_guard_attack(ctx->stream, n, View{ i->x, i->y, i->z, (long long)i->stride },
		                                   View{ o->x, o->y, o->z, (long long)o->stride }, guard6, iters, seed, best_ratio, d);
	if (e == hipSuccess) e = hipMemcpyAsync(counters3, d, 24, hipMemcpyDeviceToHost, ctx->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
	(void)hipFree(d);
	if (e != hipSuccess) return fail(DJB_ERR_HIP, "djb_error: guard attack: %s", hipGetErrorString(e));
	return DJB_OK;
}
DJB_ABI_CATCH

// ---------------------------------------------------------------- synthetic workloads
djb_status djb_gen_directions(djb_ctx *ctx, int64_t n, uint32_t seed, uint64_t start, const djb_vec3_view *out)
try {
	if (is_cpu(ctx) && out) return n <= 0 ? DJB_OK : djbcpu::gen_directions(ctx, n, seed, start, out);
	djb_status st = check_call(ctx, nullptr, n, DJB_MEM_DEVICE);
	if (st != DJB_OK) return st;
	std::lock_guard<std::recursive_mutex> call_lock(ctx->call_mu);
	if (!Staged::valid(out)) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null vec3 view");
	HIP_TRY(djbk::launch_gen_directions(ctx->stream, n, seed, start, View{ out->x, out->y, out->z, (long long)out->stride }));
	return DJB_OK;
}
DJB_ABI_CATCH
djb_status djb_gen_uniforms(djb_ctx *ctx, int64_t n, uint32_t seed, uint64_t start, float *out)
try {
	if (is_cpu(ctx)) return n <= 0 ? DJB_OK : djbcpu::gen_uniforms(ctx, n, seed, start, out);
	djb_status st = check_call(ctx, nullptr, n, DJB_MEM_DEVICE);
	if (st != DJB_OK) return st;
	std::lock_guard<std::recursive_mutex> call_lock(ctx->call_mu);
	if (!out) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null output array");
	HIP_TRY(djbk::launch_gen_uniforms(ctx->stream, n, seed, start, out));
	return DJB_OK;
}
DJB_ABI_CATCH
djb_status djb_selftest_guarded_math(djb_ctx *ctx, int64_t n, uint32_t seed, unsigned long long *counters12)
try {
	if (is_cpu(ctx)) return fail(DJB_ERR_NOT_IMPLEMENTED, "djb_error: this diagnostic needs a GPU context");
	djb_status st = check_call(ctx, nullptr, n, DJB_MEM_DEVICE);
	if (st != DJB_OK) return st;
	std::lock_guard<std::recursive_mutex> call_lock(ctx->call_mu);
	if (!counters12) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
	unsigned long long *d = nullptr;
	HIP_TRY(hipMalloc((void **)&d, 96));
	hipError_t e = hipMemsetAsync(d, 0, 96, ctx->stream);
	if (e == hipSuccess) e = djbk::launch_guard_selftest(ctx->stream, n, seed, d);
	if (e == hipSuccess) e = hipMemcpyAsync(counters12, d, 96, hipMemcpyDeviceToHost, ctx->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
	(void)hipFree(d);
	if (e != hipSuccess) return fail(DJB_ERR_HIP, "djb_error: selftest: %s", hipGetErrorString(e));
	return DJB_OK;
}
DJB_ABI_CATCH

djb_status djb_selftest_fast_trig(djb_ctx *ctx, int64_t n, int mode, uint32_t first, uint32_t seed, unsigned long long *counters4)
try {
	if (is_cpu(ctx)) return fail(DJB_ERR_NOT_IMPLEMENTED, "djb_error: this diagnostic needs a GPU context");
	djb_status st = check_call(ctx, nullptr, n, DJB_MEM_DEVICE);
	if (st != DJB_OK) return st;
	std::lock_guard<std::recursive_mutex> call_lock(ctx->call_mu);
	if (!counters4 || (mode < 0 || mode > 9)) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument / unknown mode");
	unsigned long long *d = nullptr;
	HIP_TRY(hipMalloc((void **)&d, 32));
	hipError_t e = hipMemsetAsync(d, 0, 32, ctx->stream);
	if (e == hipSuccess) e = djbk::launch_fast_trig_selftest(ctx->stream, n, mode, first, seed, d);
	if (e == hipSuccess) e = hipMemcpyAsync(counters4, d, 32, hipMemcpyDeviceToHost, ctx->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
	(void)hipFree(d);
	if (e != hipSuccess) return fail(DJB_ERR_HIP, "djb_error: selftest: %s", hipGetErrorString(e));
	return DJB_OK;
}
DJB_ABI_CATCH

djb_status djb_selftest_model_fast(djb_ctx *ctx, const djb_brdf *b, int64_t n, uint32_t seed, uint32_t first, unsigned long long *counters6)
try {
	if (is_cpu(ctx)) return fail(DJB_ERR_NOT_IMPLEMENTED, "djb_error: this diagnostic needs a GPU context");
	if (!b) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null brdf");
	djb_status st = check_call(ctx, b, n, DJB_MEM_DEVICE);
	if (st != DJB_OK) return st;
	std::lock_guard<std::recursive_mutex> call_lock(ctx->call_mu);
	if (!counters6 || (b->dev.kind != DJB_KIND_SGD && b->dev.kind != DJB_KIND_ABC))
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: djb_selftest_model_fast needs an sgd or abc brdf");
	unsigned long long *d = nullptr;
	HIP_TRY(hipMalloc((void **)&d, 48));
	hipError_t e = hipMemsetAsync(d, 0, 48, ctx->stream);
	if (e == hipSuccess) e = djbk::launch_model_fast_selftest(ctx->stream, b->dev, n, seed, first, d);
	if (e == hipSuccess) e = hipMemcpyAsync(counters6, d, 48, hipMemcpyDeviceToHost, ctx->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
	(void)hipFree(d);
	if (e != hipSuccess) return fail(DJB_ERR_HIP, "djb_error: selftest: %s", hipGetErrorString(e));
	return DJB_OK;
}
DJB_ABI_CATCH

djb_status djb_selftest_contract(djb_ctx *ctx, const djb_brdf *b, const djb_params *params, int64_t n, uint32_t seed, int family,
                                 float *max_rel2, unsigned long long *counters4)
try {
	if (is_cpu(ctx)) return fail(DJB_ERR_NOT_IMPLEMENTED, "djb_error: this diagnostic needs a GPU context");
	if (!b) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null brdf");
	djb_status st = check_call(ctx, b, n, DJB_MEM_DEVICE);
	if (st != DJB_OK) return st;
	std::lock_guard<std::recursive_mutex> call_lock(ctx->call_mu);
	if (!max_rel2 || !counters4) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
	Params p;
	if ((st = device_params(params, &p, b->dev.kind)) != DJB_OK) return st;
	const double *model_host = b->model_host.empty() ? nullptr : b->model_host.data();
	if (!djbk::contract_supported(b->dev, p, model_host))
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: brdf / params outside the domain of the contract-mode fast path");
	unsigned char *d = nullptr;
	HIP_TRY(hipMalloc((void **)&d, 64));
	hipError_t e = hipMemsetAsync(d, 0, 64, ctx->stream);
	if (e == hipSuccess)
		e = djbk::launch_contract_selftest(ctx->stream, b->dev, p, model_host, n, seed, seed ^ 0x9e3779b9u, 0ull, family,
		                                   (unsigned int *)d, (unsigned long long *)(d + 16));
	unsigned char h[64];
	if (e == hipSuccess) e = hipMemcpyAsync(h, d, 64, hipMemcpyDeviceToHost, ctx->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
	(void)hipFree(d);
	if (e != hipSuccess) return fail(DJB_ERR_HIP, "djb_error: contract selftest: %s", hipGetErrorString(e));
	memcpy(max_rel2, h, 8);
	memcpy(counters4, h + 16, 32);
	return DJB_OK;
}
DJB_ABI_CATCH

djb_status djb_selftest_contract_sample(djb_ctx *ctx, const djb_brdf *b, const djb_params *params, int64_t n, uint32_t seed, int family,
                                        float *max_abs2, unsigned long long *counters4)
try {
	float *max_abs1 = max_abs2;
	if (is_cpu(ctx)) return fail(DJB_ERR_NOT_IMPLEMENTED, "djb_error: this diagnostic needs a GPU context");
	if (!b) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null brdf");
	djb_status st = check_call(ctx, b, n, DJB_MEM_DEVICE);
	if (st != DJB_OK) return st;
	std::lock_guard<std::recursive_mutex> call_lock(ctx->call_mu);
	if (!max_abs1 || !counters4) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
	Params p;
	if ((st = device_params(params, &p, b->dev.kind)) != DJB_OK) return st;
	if (!djbk::sample_contract_supported(b->dev, p))
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: brdf / params outside the domain of the contract-mode sampler");
	unsigned char *d = nullptr;
	HIP_TRY(hipMalloc((void **)&d, 64));
	hipError_t e = hipMemsetAsync(d, 0, 64, ctx->stream);
	if (e == hipSuccess)
		e = djbk::launch_sample_contract_selftest(ctx->stream, b->dev, p, n, seed, 0ull, family, (unsigned int *)d, (unsigned long long *)(d + 16));
	unsigned char h[64];
	if (e == hipSuccess) e = hipMemcpyAsync(h, d, 64, hipMemcpyDeviceToHost, ctx->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
	(void)hipFree(d);
	if (e != hipSuccess) return fail(DJB_ERR_HIP, "djb_error: contract sampler selftest: %s", hipGetErrorString(e));
	memcpy(max_abs1, h, 8);
	memcpy(counters4, h + 16, 32);
	return DJB_OK;
}
DJB_ABI_CATCH

djb_status djb_contract_sample_attack(djb_ctx *ctx, const djb_brdf *b, const djb_params *params, int64_t n, float *u1, float *u2,
                                      const djb_vec3_view *o, int iters, uint32_t seed, float *best, unsigned long long *counters3)
try {
	if (is_cpu(ctx)) return fail(DJB_ERR_NOT_IMPLEMENTED, "djb_error: this diagnostic needs a GPU context");
	if (!b) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null brdf");
	djb_status st = check_call(ctx, b, n, DJB_MEM_DEVICE);
	if (st != DJB_OK) return st;
	std::lock_guard<std::recursive_mutex> call_lock(ctx->call_mu);
	if (!Staged::valid(o) || !u1 || !u2 || !best || !counters3 || iters < 0) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: invalid argument");
	Params p;
	if ((st = device_params(params, &p, b->dev.kind)) != DJB_OK) return st;
	if (!djbk::sample_contract_supported(b->dev, p))
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: brdf / params outside the domain of the contract-mode sampler");
	unsigned long long *d = nullptr;
	HIP_TRY(hipMalloc((void **)&d, 32));
	hipError_t e = hipMemsetAsync(d, 0, 32, ctx->stream);
	if (e == hipSuccess)
		e = djbk::launch_sample_contract_attack(ctx->stream, b->dev, p, n, u1, u2, View{ o->x, o->y, o->z, (long long)o->stride }, iters, seed, best, d);
	if (e == hipSuccess) e = hipMemcpyAsync(counters3, d, 24, hipMemcpyDeviceToHost, ctx->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
	(void)hipFree(d);
	if (e != hipSuccess) return fail(DJB_ERR_HIP, "djb_error: contract sampler attack: %s", hipGetErrorString(e));
	return DJB_OK;
}
DJB_ABI_CATCH

djb_status djb_selftest_libm(djb_ctx *ctx, int fn, int64_t n, const double *x, const double *y, double *out)
try {
	if (is_cpu(ctx)) return fail(DJB_ERR_NOT_IMPLEMENTED, "djb_error: this diagnostic needs a GPU context");
	djb_status st = check_call(ctx, nullptr, n, DJB_MEM_HOST);
	if (st != DJB_OK) return st;
	std::lock_guard<std::recursive_mutex> call_lock(ctx->call_mu);
	if (fn < 0 || fn > 11 || !x || !y || !out) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: invalid selftest arguments");
	if (n == 0) return DJB_OK;
	const size_t nb = sizeof(double) * (size_t)n;
	double *d = nullptr;
	HIP_TRY(hipMalloc((void **)&d, 3 * nb));
	hipError_t e = hipMemcpy(d, x, nb, hipMemcpyHostToDevice);
	if (e == hipSuccess) e = hipMemcpy(d + n, y, nb, hipMemcpyHostToDevice);
	if (e == hipSuccess) e = djbk::launch_libm_probe(ctx->stream, fn, n, d, d + n, d + 2 * n);
	if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
	if (e == hipSuccess) e = hipMemcpy(out, d + 2 * n, nb, hipMemcpyDeviceToHost);
	(void)hipFree(d);
	if (e != hipSuccess) return fail(DJB_ERR_HIP, "djb_error: selftest: %s", hipGetErrorString(e));
	return DJB_OK;
}
DJB_ABI_CATCH

static_assert(djbdev::TRIG_SITES == DJB_TRIG_SITES && djbdev::TRIG_DOUBLE == DJB_TRIG_DOUBLE && djbdev::TRIG_DOUBLE_SITES == DJB_TRIG_DOUBLE_SITES, "djb_hip.h and djb_device.hpp number the trig sites differently");
static bool trig_site_valid(int fn)
{
	return (fn >= 0 && fn < DJB_TRIG_SITES) || (fn >= DJB_TRIG_DOUBLE && fn < DJB_TRIG_DOUBLE + DJB_TRIG_DOUBLE_SITES);
}
djb_status djb_selftest_trig_sweep(djb_ctx *ctx, int fn, int host_fn, uint32_t first_bits, int64_t count, int threads,
                                   unsigned long long *n_bad, uint32_t *bad3, int cap)
try {
	if (is_cpu(ctx)) return fail(DJB_ERR_NOT_IMPLEMENTED, "djb_error: this diagnostic needs a GPU context");
	djb_status st = check_call(ctx, nullptr, count, DJB_MEM_HOST);
	if (st != DJB_OK) return st;
	if (!trig_site_valid(fn) || !trig_site_valid(host_fn) || (fn >= DJB_TRIG_DOUBLE) != (host_fn >= DJB_TRIG_DOUBLE) || !n_bad ||
	    cap < 0 || (cap > 0 && !bad3) || count > ((int64_t)1 << 32) - (int64_t)first_bits)
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: invalid selftest arguments");
	*n_bad = 0;
	if (count == 0) return DJB_OK;
	const size_t nb = (fn >= DJB_TRIG_DOUBLE ? sizeof(double) : sizeof(float)) * (size_t)count;
	std::vector<char> host(nb);
	{
		std::lock_guard<std::recursive_mutex> call_lock(ctx->call_mu);
		void *d = nullptr;
		HIP_TRY(hipMalloc(&d, nb));
		hipError_t e = djbk::launch_trig_sweep(ctx->stream, fn, first_bits, count, d);
		if (e == hipSuccess) e = hipMemcpyAsync(host.data(), d, nb, hipMemcpyDeviceToHost, ctx->stream);
		if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
		(void)hipFree(d);
		if (e != hipSuccess) return fail(DJB_ERR_HIP, "djb_error: selftest: %s", hipGetErrorString(e));
	}
	*n_bad = djbcpu::trig_sweep_compare(host_fn, first_bits, count, host.data(), threads, bad3, cap);
	return DJB_OK;
}
DJB_ABI_CATCH

djb_status djb_histogram_xy(djb_ctx *ctx, int64_t n, const djb_vec3_view *v, int bins, unsigned long long *counts)
try {
	if (is_cpu(ctx)) return djbcpu::histogram_xy(ctx, n, v, bins, counts);
	djb_status st = check_call(ctx, nullptr, n, DJB_MEM_DEVICE);
	if (st != DJB_OK) return st;
	std::lock_guard<std::recursive_mutex> call_lock(ctx->call_mu);
	if (!Staged::valid(v) || !counts || bins < 1 || bins > 128)
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: invalid histogram arguments");
	HIP_TRY(djbk::launch_histogram_xy(ctx->stream, n, View{ v->x, v->y, v->z, (long long)v->stride }, bins, counts));
	return DJB_OK;
}
DJB_ABI_CATCH

} // extern "C"
