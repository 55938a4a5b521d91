// Shared by the 3D launch units (gmg_launch3d.hip, gmg_fused3d.hip, gmg_patchsolve.hip): making a level's ghost planes current
// around a kernel launch -- pack, face exchange (or the interior / boundary split around it), coarse/fine ghosts -- and shipping
// restricted blocks to the ranks that hold the parents. Templates on the patch size, instantiated where they are used.
#pragma once
#include "gmg_internal.hpp"

namespace tei
{
// make every ghost plane of `u` current: remote same-level faces (pack -> exchange -> ghost slots
// [0, nremote)), then the coarse/fine planes. Replaces SchurHelper.h:145-150 updateInterfaceDist.
// `ps`: the iterate is u + P(ps->coarse) (never stored): the faces are packed with the correction added.
// may_push: the exchange that follows is on the solver stream -- with the direct-store transport the pack kernel then stores the
// layers into the receivers' ghost slots itself and raises their flags (PackPush); returns true when it did (the caller finishes
// with pushFinish instead of an exchange of the send buffer)
template <int N> bool packFaces(te_gmg *g, LevelHost &L, const double *u, const ProlongSrc *ps, bool may_push = false)
{
	const bool push = may_push && g->push.on && L.push_faces && !g->recording && L.push_face_dst[0].p;
	Timed      t(g, push ? KC_EXCHANGE : KC_PACK, (size_t) L.nremote * L.nf);
	const dim3 grid(L.nremote), blk(N * N < 256 ? N * N : 256);
	PackPush   pp;
	if (push) {
		pushBegin(g, L, 1);
		pp.dst    = L.push_face_dst[L.push_par].p;
		pp.flags  = L.push_face_flags.p;
		pp.nflags = (int) L.push_face_flags.n;
		pp.epoch  = L.push_ep;
		pp.raise  = pushRaiseValue(g, L.push_ep);
		pp.done   = L.push_done.p;
		pp.err    = g->push.err;
		pp.err_host = g->push.err_host;
	}
	if (L.pack_f6) { // the iterate exists only as its face layers
		ProlongSrc none{nullptr, nullptr, nullptr};
		hipLaunchKernelGGL(k_pack_faces6_3d<N>, grid, blk, 0, g->stream, L.send_faces.p, L.pack_f6, ps ? *ps : none, L.sendbuf.p, L.f6Off(), pp);
	} else if (ps)
		hipLaunchKernelGGL(k_pack_faces_prolong3d<N>, grid, blk, 0, g->stream, L.send_faces.p, u, *ps, L.sendbuf.p, pp);
	else
		hipLaunchKernelGGL(k_pack_faces3d<N>, grid, blk, 0, g->stream, L.send_faces.p, u, L.sendbuf.p, pp);
	return push;
}

// ghost planes of the coarse/fine faces from the current iterate (u, or its face layers L.pack_f6 when it was never stored)
template <int N> void cfGhosts(te_gmg *g, LevelHost &L, const double *u, const ProlongSrc *ps)
{
	Timed      t(g, KC_CFGHOST, (size_t) L.ncf * L.nf);
	const dim3 grid(L.ncf), blk(N * N < 256 ? N * N : 256);
	if (L.pack_f6) {
		if (ps)
			hipLaunchKernelGGL((k_cf_ghost6_3d<N, true>), grid, blk, 0, g->stream, L.cf_desc.p, L.cf_slots.p, L.pack_f6, *ps, L.ghostCur(), L.f6Off());
		else
			hipLaunchKernelGGL((k_cf_ghost6_3d<N, false>), grid, blk, 0, g->stream, L.cf_desc.p, L.cf_slots.p, L.pack_f6, ProlongSrc(), L.ghostCur(), L.f6Off());
	} else if (ps)
		hipLaunchKernelGGL(k_cf_ghost_prolong3d<N>, grid, blk, 0, g->stream, L.cf_desc.p, L.cf_slots.p, u, *ps, L.ghostCur());
	else
		hipLaunchKernelGGL(k_cf_ghost3d<N>, grid, blk, 0, g->stream, L.cf_desc.p, L.cf_slots.p, u, L.ghostCur());
}

template <int N> int prepareGhosts(te_gmg *g, LevelHost &L, const double *u, const ProlongSrc *ps = nullptr)
{
	if (L.patch_local) return TE_OK; // the patch operator reads no neighbour
	L.ghost_has_v = false;
	if (L.nremote > 0) {
		// the face layers of an iterate that exists only as such already sit in send order (LevelHost::f6off): sent from there
		const bool direct = L.pack_f6 && !ps && L.f6Off();
		const bool pushed = !direct && packFaces<N>(g, L, u, ps, true);
		int        rc     = pushed ? pushFinish(g, L, 1, g->stream) : faceExchange(g, L, direct ? L.pack_f6 : L.sendbuf.p);
		if (rc) return rc;
	}
	if (L.ncf == 0) return TE_OK;
	cfGhosts<N>(g, L, u, ps);
	return TE_OK;
}

// Run `launch(subset)` over all patches of the level with current ghosts. With off-rank neighbours the
// exchange goes to the communication stream and the interior patches (no ghost-slot face) are computed
// underneath it; the boundary patches follow once the receive has landed. (north star: "ghost-cell
// exchange ... overlapped with interior smoothing")
template <int N, class F> int withGhosts(te_gmg *g, LevelHost &L, const double *u, F launch_, const double *xf_in = nullptr,
                                         double *xf_out = nullptr, const ProlongSrc *ps = nullptr)
{
	auto launch = [&](LevelDev D) {
		D.xf     = xf_in;
		D.xf_out = xf_out;
		launch_(D);
	};
	// (levels with few local patches: nothing worth hiding under the exchange, and the second stream and its two
	// events only add host calls and latency)
	// TE_OVERLAP_MIN (tests set 0 so that their small levels take the overlapped path). The split costs a second launch -- no
	// launch is shorter than one patch march, about 30 us -- and two cross-stream event hand-overs of about 20 us each, so it
	// pays only where the interior launch is much longer than that: more local patches than the chip holds workgroups at once
	// (3 x 256). Measured per rank with the exchanges in loop-back (tools/mr8_budget.py): at 512 local patches (512^3 on eight
	// ranks) the cycle is 525 us with the split and 475 us without it.
	L.ghost_has_v = false;
	const int mode = L.overlap_mode >= 0 ? L.overlap_mode : (L.P < g->cfg.num(O_OVERLAP_MIN, 768) ? 0 : (g->cfg.num(O_OVERLAP_MODE, 1) == 2 ? 2 : 1));
	if (L.patch_local || g->recording || L.nremote == 0 || !g->overlap || L.n_int == 0 || mode == 0) {
		int rc = prepareGhosts<N>(g, L, u, ps);
		if (rc) return rc;
		launch(L.dev());
		return TE_OK;
	}
	if (mode == 2) {
		// the interior patches go to the second stream (they wait for nothing but what the solver stream has done so far); pack,
		// exchange, coarse/fine ghosts and the boundary patches stay on the solver stream, which then waits for the interior
		HIPCHK(hipEventRecord(g->ev_pack, g->stream));
		HIPCHK(hipStreamWaitEvent(g->comm_stream, g->ev_pack, 0));
		std::swap(g->stream, g->comm_stream); // (every launch helper launches on g->stream)
		launch(L.devPart(false));
		hipError_t e = hipEventRecord(g->ev_recv, g->stream);
		std::swap(g->stream, g->comm_stream);
		HIPCHK(e);
		const bool pushed = packFaces<N>(g, L, u, ps, true);
		int        rc     = pushed ? pushFinish(g, L, 1, g->stream) : faceExchange(g, L, L.sendbuf.p);
		if (rc) return rc;
		if (L.ncf > 0) cfGhosts<N>(g, L, u, ps);
		launch(L.devPart(true));
		HIPCHK(hipStreamWaitEvent(g->stream, g->ev_recv, 0));
		return TE_OK;
	}
	packFaces<N>(g, L, u, ps);
	HIPCHK(hipEventRecord(g->ev_pack, g->stream));
	HIPCHK(hipStreamWaitEvent(g->comm_stream, g->ev_pack, 0));
	int rc = faceExchange(g, L, L.sendbuf.p, g->comm_stream);
	if (rc) return rc;
	HIPCHK(hipEventRecord(g->ev_recv, g->comm_stream));
	launch(L.devPart(false)); // interior, concurrent with the exchange
	HIPCHK(hipStreamWaitEvent(g->stream, g->ev_recv, 0));
	if (L.ncf > 0) cfGhosts<N>(g, L, u, ps);
	launch(L.devPart(true)); // boundary
	return TE_OK;
}

// The restricted blocks of this rank's patches to the other ranks (the kernels before this have written them into the local
// coarse patches or into upbuf), the other ranks' blocks into the local coarse patches. repl_up (the coarse level lives on
// every rank): the finished octants are copied out of the coarse patches first -- one copy, sent to everybody.
template <int N> int shipRestricted(te_gmg *g, LevelHost &L, double *coarse)
{
	// every rank's blocks are whole coarse patches in one run of the coarse vector: run to run, in place
	if (L.repl_up && L.repl_direct && !g->cfg.has(O_REPL_BLOCKS)) {
		if (g->push.on && L.push_blocks && !g->recording) return pushExchange(g, L, 2, coarse);
		return doExchange(g, 2, L.tx_direct, coarse, coarse);
	}
	if (L.repl_up && L.n_up > 0) {
		Timed t(g, KC_PACK, (size_t) L.n_up * L.nc / 8);
		hipLaunchKernelGGL(k_prolong_pack3d<N>, dim3(L.n_up), dim3(256), 0, g->stream, L.bc_desc.p, L.up_off.p, coarse, L.upbuf.p);
	}
	int rc = doExchange(g, 2, L.tx_up, L.upbuf.p, L.downbuf.p);
	if (rc) return rc;
	if (L.n_down > 0) {
		Timed t(g, KC_PACK, (size_t) L.n_down * L.nc / 8);
		hipLaunchKernelGGL(k_restrict_unpack3d<N>, dim3(L.n_down), dim3(256), 0, g->stream, L.down_desc.p, L.down_off.p,
		                   L.downbuf.p, coarse);
	}
	return TE_OK;
}
} // namespace tei
