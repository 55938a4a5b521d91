// Level tables on the device, solver and vector life cycle, options, profiling, Init kernels (see gmg_internal.hpp).
#include "gmg_internal.hpp"

namespace tei
{
const char *kclassName[KC_COUNT] = {"stencil_apply", "stencil_resid", "stencil_jacobi", "stencil_rbgs",
                                    "cf_ghost", "restrict", "prolong_add", "patch_rhs", "dst_axis",
                                    "vecop", "reduce", "pack", "exchange", "stencil_rbgs_zero", "resid_restrict", "patch_solve_mfma", "stencil_rbgs_prolong",
                                    "stencil_rbgs_slabs", "stencil_slabs", "patch_solve_3pass", "rbgs_zero_resid_restrict",
                                    "restrict_fixup", "rbgs_resweep_prolong", "rbgs_zero_resid_restrict_faces",
                                    "rbgs_resweep_prolong_fcorr", "rbgs_zero_resid_restrict_faces_fcorr", "fcorr_gather", "patch_solve_mfma_faces",
                                    "bicg_update", "bicg_s", "bicg_p", "stencil_apply_dot", "patch_bcgs"};
const char *optName[O_COUNT] = {"TE_2D_SIMPLE", "TE_2D_NO_MFMA", "TE_2D_NO_PF", "TE_2D_NO_MR_FUSE", "TE_2D_TPB", "TE_NO_FUSE2", "TE_NO_FUSE3",
                                "TE_NO_FUSE3_CF", "TE_NO_CFP", "TE_NO_XF", "TE_NO_FCORR", "TE_NO_FCORR_CF", "TE_NO_GTAB", "TE_NO_OVERLAP",
                                "TE_OVERLAP_MIN", "TE_NO_PS_FACES", "TE_PS_MODE", "TE_PS_SLOW", "TE_RBGS_NOSLAB", "TE_ZS_FORCE", "TE_NO_ZS8",
                                "TE_RESWEEP_V", "TE_EXCHANGE_TIMEOUT", "TE_NO_VERIFY", "TE_RCCL_LOOPBACK", "TE_ZR_AHEAD", "TE_NO_BICG_FUSE", "TE_POST_EXCHANGE", "TE_REPL_BLOCKS", "TE_PACK_FACES", "TE_OVERLAP_MODE", "TE_PUSH_TIMEOUT", "TE_NO_BICG_XF", "TE_PUSH_FAULT", "TE_2D_NO_FOLD", "TE_2D_NO_SYM", "TE_PUSH_NONFATAL", "TE_PS_NO_HALF", "TE_PS_HALF_MAX", "TE_NO_GTAB2", "TE_NO_RS6_CF", "TE_NO_RS6_FIXUP", "TE_NO_CFP59"};

void drainEvents(te_gmg *g)
{
	if (g->ev_used == 0) return;
	(void) hipStreamSynchronize(g->stream);
	for (size_t i = 0; i < g->ev_used; i++) {
		float ms = 0;
		if (g->ev_pool[i].valid && hipEventElapsedTime(&ms, g->ev_pool[i].a, g->ev_pool[i].b) == hipSuccess) {
			g->calls[g->ev_pool[i].kc]++;
			g->total_ms[g->ev_pool[i].kc] += ms;
		}
	}
	g->ev_used = 0;
}

void Cfg::fromEnv()
{
	for (int o = 0; o < O_COUNT; o++) set(o, getenv(optName[o]));
}

// DftPatchSolver.h:237-289 (row-major: y_i = sum_j M[i*n+j] x_j)
void transformMatrix(int type, int n, double *m)
{
	for (int i = 0; i < n * n; i++) m[i] = 0.0;
	switch (type) {
		case 0: // DCT-II
			for (int i = 0; i < n; i++)
				for (int j = 0; j < n; j++) m[i * n + j] = cos(M_PI / n * (i * (j + 0.5)));
			break;
		case 1: // DCT-III
			for (int i = 0; i < n; i++) {
				m[i * n] = 0.5;
				for (int j = 1; j < n; j++) m[i * n + j] = cos(M_PI / n * ((i + 0.5) * j));
			}
			break;
		case 2: // DCT-IV
			for (int i = 0; i < n; i++)
				for (int j = 0; j < n; j++) m[i * n + j] = cos(M_PI / n * ((i + 0.5) * (j + 0.5)));
			break;
		case 3: // DST-II
			for (int i = 0; i < n; i++)
				for (int j = 0; j < n; j++) m[i * n + j] = sin(M_PI / n * ((i + 1) * (j + 0.5)));
			break;
		case 4: // DST-III
			for (int i = 0; i < n; i++) {
				for (int j = 0; j < n - 1; j++) m[i * n + j] = sin(M_PI / n * ((i + 0.5) * (j + 1)));
				m[i * n + n - 1] = (i & 1) ? -0.5 : 0.5;
			}
			break;
		default: // DST-IV
			for (int i = 0; i < n; i++)
				for (int j = 0; j < n; j++) m[i * n + j] = sin(M_PI / n * ((i + 0.5) * (j + 0.5)));
			break;
	}
}

// per-peer ranges of a list of (peer, count) items that is already sorted by peer
void rangesByPeer(const std::vector<std::pair<int, int64_t>> &items, std::vector<int32_t> &peers,
                  std::vector<int64_t> &off, std::vector<int64_t> &cnt)
{
	int64_t pos = 0;
	for (auto &it : items) {
		if (peers.empty() || peers.back() != it.first) {
			peers.push_back(it.first);
			off.push_back(pos);
			cnt.push_back(0);
		}
		cnt.back() += it.second;
		pos += it.second;
	}
}

// merge the send-side and receive-side peer lists of one exchange into one ExPlan
ExPlan mergePlan(const std::vector<std::pair<int, int64_t>> &sends, const std::vector<std::pair<int, int64_t>> &recvs)
{
	std::vector<int32_t> sp, rp;
	std::vector<int64_t> so, sc, ro, rc;
	rangesByPeer(sends, sp, so, sc);
	rangesByPeer(recvs, rp, ro, rc);
	std::map<int, std::array<int64_t, 4>> m;
	for (size_t i = 0; i < sp.size(); i++) m[sp[i]] = {so[i], sc[i], 0, 0};
	for (size_t i = 0; i < rp.size(); i++) {
		auto &e = m[rp[i]];
		e[2]    = ro[i];
		e[3]    = rc[i];
	}
	ExPlan pl;
	for (auto &kv : m) {
		pl.peers.push_back(kv.first);
		pl.send_off.push_back(kv.second[0]);
		pl.send_cnt.push_back(kv.second[1]);
		pl.recv_off.push_back(kv.second[2]);
		pl.recv_cnt.push_back(kv.second[3]);
	}
	return pl;
}

int buildLevel(te_gmg *g, const Hierarchy &H, int li)
{
	const Level &lv = H.levels[li];
	const int n = lv.n, D = lv.dim;
	if (D == 3 && n != 4 && n != 8 && n != 16 && n != 32)
		return te::fail(TE_EUNSUPPORTED, "te_gmg_create: 3D patches must have n = 4, 8, 16 or 32 cells per axis");
	if (D == 2 && (n < 4 || (n & 1))) return te::fail(TE_EUNSUPPORTED, "te_gmg_create: 2D patches need an even n >= 4");
	auto L = std::make_unique<LevelHost>();
	L->dim = D;
	L->n   = n;
	L->P   = lv.P;
	L->P_global = lv.P_global;
	L->index    = li;
	L->replicated = lv.replicated;
	L->gathered = lv.replicated || (H.nranks > 1 && std::all_of(lv.g_rank.begin(), lv.g_rank.end(), [&](int32_t r) { return r == lv.g_rank[0]; }));
	L->nc  = (D == 3) ? (size_t) n * n * n : (size_t) n * n;
	L->nf  = (D == 3) ? (size_t) n * n : (size_t) n;
	const int P = lv.P, NS = 2 * D, NCH = 1 << D, NQ = 1 << (D - 1), me = H.rank;

	// ---- remote same-level faces: canonical order = (peer, receiving patch (global), receiving side),
	// which both ends can compute from the global tables
	// A receiving (patch, side, q) gets ONE ghost slot holding the sender's facing layer: the ghost values
	// themselves on a same-level face, raw neighbour cells for k_cf_ghost on a coarse/fine face (q = which
	// of the finer neighbours; the coarse side of a coarse/fine face receives one slot per fine neighbour).
	struct RFace {
		int peer, key_patch, key_side, key_q, p, s, nb;
		bool operator<(const RFace &o) const
		{
			return std::tie(peer, key_patch, key_side, key_q) < std::tie(o.peer, o.key_patch, o.key_side, o.key_q);
		}
	};
	std::vector<RFace> recvs, sends;
	for (int p = 0; p < P; p++) {
		const int gp = lv.l2g[p];
		for (int s = 0; s < NS; s++) {
			const size_t gf   = (size_t) gp * NS + s;
			const int    kind = lv.g_nbr_kind[gf];
			if (kind == NBR_NONE) continue;
			for (int q = 0; q < NQ; q++) {
				const int nb = lv.g_nbr[gf * 4 + q];
				if (nb < 0 || lv.g_rank[nb] == me) continue;
				recvs.push_back({lv.g_rank[nb], gp, s, q, p, s, nb});
				// what the neighbour files my layer under: its own (patch, side) and, when it is the coarse
				// side, my position among its fine neighbours = my quadrant on its face
				const int their_q = (kind == NBR_COARSE) ? lv.g_nbr_orth[gf] : 0;
				sends.push_back({lv.g_rank[nb], nb, s ^ 1, their_q, p, s, nb});
			}
		}
	}
	std::sort(recvs.begin(), recvs.end());
	std::sort(sends.begin(), sends.end());
	const int                           nremote = (int) recvs.size();
	std::map<std::tuple<int, int, int>, int> remote_slot; // (p, s, q) -> ghost slot
	for (int i = 0; i < nremote; i++) remote_slot[std::make_tuple(recvs[i].p, recvs[i].s, recvs[i].key_q)] = i;
	{
		std::vector<std::pair<int, int64_t>> si, ri;
		std::vector<int32_t>                 sf;
		for (auto &f : sends) {
			si.emplace_back(f.peer, (int64_t) L->nf);
			sf.push_back(f.p);
			sf.push_back(f.s);
		}
		for (auto &f : recvs) ri.emplace_back(f.peer, (int64_t) L->nf);
		L->fx      = mergePlan(si, ri);
		L->nremote = nremote;
		int rc0;
		if ((rc0 = L->send_faces.upload(sf)) || (rc0 = L->sendbuf.alloc((size_t) std::max(nremote, 1) * L->nf))) return rc0;
		if (D == 3 && nremote > 0) { // the place of every face layer in f6buf: sent layers first, in send order (see LevelHost::f6off)
			std::vector<int32_t> off((size_t) P * NS, -1);
			bool                 once = true;
			for (size_t i = 0; i < sends.size() && once; i++) {
				int32_t &o = off[(size_t) sends[i].p * NS + sends[i].s];
				once       = (o < 0);
				o          = (int32_t) i;
			}
			if (once) {
				int32_t next = (int32_t) sends.size();
				for (auto &o : off)
					if (o < 0) o = next++;
				if ((rc0 = L->f6off.upload(off))) return rc0;
			}
		}
	}

	std::vector<int32_t> fk(P * NS), fs(P * NS, -1), cfd, cfs, plan(P, 0);
	std::vector<double>  kadj(P * NS, 0.0), rh2(P * 3), cellvol(P);
	L->patch_vol.assign(P, 0.0);
	std::map<int, int>   plan_of_key;
	std::vector<int>     keys;
	int                  nslots = nremote;
	for (int p = 0; p < P; p++) {
		const int gp  = lv.l2g[p];
		int       key = 0;
		rh2[p * 3 + 2] = 0.0;
		double cv = 1.0, pv = 1.0; // Domain.h:270-272 (patch_sum *= spacings[i]), :242-245
		for (int a = 0; a < D; a++) {
			double h       = lv.g_lengths[(size_t) gp * D + a] / n;
			rh2[p * 3 + a] = 1.0 / (h * h);
			cv *= h;
			pv *= h * n;
		}
		cellvol[p]      = cv;
		L->patch_vol[p] = pv;
		for (int s = 0; s < NS; s++) {
			const size_t gf   = (size_t) gp * NS + s;
			const int    kind = lv.g_nbr_kind[gf];
			if (kind == NBR_NONE) {
				fk[p * NS + s]   = H.neumann ? FACE_NEUMANN : FACE_DIRICHLET;
				kadj[p * NS + s] = H.neumann ? -1.0 : 1.0;
				if (H.neumann) key |= 1 << s;
			} else if (kind == NBR_NORMAL) {
				const int nb = lv.g_nbr[gf * 4];
				if (lv.g_rank[nb] == me) {
					fk[p * NS + s] = FACE_LOCAL;
					fs[p * NS + s] = lv.g_local[nb];
				} else { // the neighbour's face cells arrive in a ghost slot; diagonal unchanged
					fk[p * NS + s] = FACE_GHOST;
					fs[p * NS + s] = remote_slot.at(std::make_tuple(p, s, 0));
				}
			} else {
				fk[p * NS + s]   = FACE_GHOST;
				fs[p * NS + s]   = nslots;
				kadj[p * NS + s] = (kind == NBR_COARSE) ? (D == 3 ? -5.0 / 6.0 : -2.0 / 3.0) : 1.0 / 3.0;
				cfd.push_back(p);
				cfd.push_back(s);
				cfd.push_back(kind);
				cfd.push_back(lv.g_nbr_orth[gf]);
				for (int q = 0; q < 4; q++) { // local patch index, or -(slot+2) of the raw layer received for it
					int nb = (q < NQ) ? lv.g_nbr[gf * 4 + q] : -1;
					if (nb >= 0 && lv.g_rank[nb] != me)
						cfd.push_back(-(remote_slot.at(std::make_tuple(p, s, q)) + 2));
					else
						cfd.push_back(nb >= 0 ? lv.g_local[nb] : -1);
				}
				cfs.push_back(nslots);
				nslots++;
			}
		}
		auto it = plan_of_key.find(key);
		if (it == plan_of_key.end()) {
			plan_of_key[key] = (int) keys.size();
			plan[p]          = (int) keys.size();
			keys.push_back(key);
		} else {
			plan[p] = it->second;
		}
	}
	L->nslots = nslots;
	L->lds2d  = (D == 2 && n <= 64 && n % 2 == 0 && !g->cfg.has(O_2D_SIMPLE));
	// see LevelHost::fuse2_ok: a global fact only. (Refined levels qualify: patches that copy through and
	// coarse/fine faces -- whose ghost slots carry the interpolated value -- are handled by both kernels.)
	L->fuse2_ok = (D == 3 && li + 1 < (int) H.levels.size() && lv.P_global >= 256); // (TE_NO_FUSE2 is looked at where the path is chosen)
	L->ncf    = (int) cfs.size();
	int rc;
	{
		std::vector<int32_t> ord, bnd;
		for (int p = 0; p < P; p++) {
			bool b = false;
			for (int s = 0; s < NS; s++) b |= (fk[p * NS + s] == FACE_GHOST);
			(b ? bnd : ord).push_back(p);
		}
		L->n_int = (int) ord.size();
		L->n_bnd = (int) bnd.size();
		ord.insert(ord.end(), bnd.begin(), bnd.end());
		if ((rc = L->order.upload(ord))) return rc;
	}
	if (D == 3 && ((rc = L->xfbuf[0].alloc((size_t) std::max(P, 1) * 2 * L->nf)) || (rc = L->xfbuf[1].alloc((size_t) std::max(P, 1) * 2 * L->nf))
	               || (rc = L->f6buf.alloc((size_t) std::max(P, 1) * 6 * L->nf))))
		return rc;
	if (D == 3 && li > 0 && L->fuse2_ok && P > 0) { // a level that can read its right-hand side with FCORR
		if ((rc = L->fcorr.alloc((size_t) P * 4 * L->nf))) return rc;
		HIPCHK(hipMemset(L->fcorr.p, 0, sizeof(double) * L->fcorr.n));
	}
	if (D == 3 && L->fuse2_ok && P > 0 && (rc = L->rs6.alloc((size_t) P * 6 * L->nf / 4))) return rc;
	{
		std::vector<double>  gs((size_t) P * 3, 0.0), gh((size_t) P * 3, 1.0);
		std::vector<int32_t> ids(P);
		for (int p = 0; p < P; p++) {
			const int gp = lv.l2g[p];
			ids[p]       = lv.g_id[gp];
			for (int a = 0; a < D; a++) {
				gs[(size_t) p * 3 + a] = lv.g_starts[(size_t) gp * D + a];
				gh[(size_t) p * 3 + a] = lv.g_lengths[(size_t) gp * D + a] / n;
			}
		}
		if ((rc = L->geom_starts.upload(gs)) || (rc = L->geom_h.upload(gh)) || (rc = L->node_ids.upload(ids))) return rc;
	}
	if ((rc = L->cellvol.upload(cellvol))) return rc;
	{
		std::vector<int32_t> fkp(fk);
		for (auto &k : fkp)
			if (k >= FACE_LOCAL) k = FACE_DIRICHLET;
		if ((rc = L->face_kind_patch.upload(fkp))) return rc;
	}
	if ((rc = L->face_kind.upload(fk)) || (rc = L->face_src.upload(fs)) || (rc = L->face_kadj.upload(kadj))
	    || (rc = L->rh2.upload(rh2)) || (rc = L->cf_desc.upload(cfd)) || (rc = L->cf_slots.upload(cfs))
	    || (rc = L->ghost.alloc((size_t) std::max(nslots, 1) * L->nf)))
		return rc;

	// patch-solve plans (FftwPatchSolver.h:93-172: transform kinds per axis, eigenvalues)
	{
		const int            np = (int) keys.size();
		std::vector<double>  mats((size_t) np * 2 * D * n * n), lam((size_t) np * D * n);
		std::vector<int32_t> zm(np, 0);
		for (int k = 0; k < np; k++) {
			const int key = keys[k];
			zm[k]         = (key == (1 << NS) - 1);
			for (int a = 0; a < D; a++) {
				bool lo = (key >> (2 * a)) & 1, hi = (key >> (2 * a + 1)) & 1;
				int  tf, ti;
				if (lo && hi) {
					tf = 0;
					ti = 1;
				} else if (lo) {
					tf = ti = 2;
				} else if (hi) {
					tf = ti = 5;
				} else {
					tf = 3;
					ti = 4;
				}
				transformMatrix(tf, n, &mats[((size_t) k * 2 * D + a) * n * n]);
				transformMatrix(ti, n, &mats[((size_t) k * 2 * D + D + a) * n * n]);
				for (int i = 0; i < n; i++) {
					double s;
					if (lo && hi)
						s = sin(i * M_PI / (2 * n));
					else if (lo || hi)
						s = sin((i + 0.5) * M_PI / (2 * n));
					else
						s = sin((i + 1) * M_PI / (2 * n));
					lam[((size_t) k * D + a) * n + i] = 4 * s * s;
				}
			}
		}
		if (D == 3 && n == 32) { // the three-pass kernels' matrices in lane order (patchsolve32.hpp matFragSource)
			std::vector<double> mf((size_t) np * 6 * 1024);
			for (int k = 0; k < np; k++)
				for (int m = 0; m < 6; m++)
					for (int e = 0; e < 16; e++)
						for (int ln = 0; ln < 64; ln++)
							mf[((size_t) k * 6 + m) * 1024 + ((size_t) (e >> 1) * 64 + ln) * 2 + (e & 1)] = mats[((size_t) k * 6 + m) * 1024 + matFragSource(m, ln, e)];
			if ((rc = L->matfrag.upload(mf))) return rc;
		}
		if (D == 3 && n == 32) { // k_ps_sym's tables: [plan][transform 6][parity 2][k-step 4][lane 64]
			std::vector<double> fs((size_t) np * PSS_FRAG, 0.0);
			bool                pure = true;
			for (int k = 0; k < np; k++)
				for (int a = 0; a < 3; a++) {
					const bool lo = (keys[k] >> (2 * a)) & 1, hi = (keys[k] >> (2 * a + 1)) & 1;
					if (lo != hi) {
						pure = false;
						continue;
					}
					const double *F = &mats[((size_t) k * 6 + a) * n * n], *G = &mats[((size_t) k * 6 + 3 + a) * n * n];
					for (int p = 0; p < 2; p++)
						for (int q = 0; q < 4; q++)
							for (int ln = 0; ln < 64; ln++) {
								const int j = ln & 15, g = ln >> 4;
								// forward: y as B operand and z as A operand take k = n = 4q + g, x as A operand k = g + 4q
								// (y comes first in the kernel: slot 0 = y, 1 = x, 2 = z)
								// inverse: x as B operand (k = m = 4q + g), y and z as A operands with k = m = g + 4q
								const int nf = (a == 0) ? g + 4 * q : 4 * q + g, mi = (a == 0) ? 4 * q + g : g + 4 * q;
								const int sf = (a == 0) ? 1 : (a == 1 ? 0 : 2);
								fs[(size_t) k * PSS_FRAG + ((sf * 2 + p) * 4 + q) * 64 + ln]      = F[(2 * j + p) * n + nf];
								// (the y inverse is the last product of the solve: its fragments carry the scale (2/N)^3 = 2^-12 of
								// DftPatchSolver.h:214 -- a power of two: the same bits as a multiplication of the result)
								fs[(size_t) k * PSS_FRAG + (((3 + a) * 2 + p) * 4 + q) * 64 + ln] = G[j * n + 2 * mi + p] * (a == 1 ? 8.0 / (32.0 * 32.0 * 32.0) : 1.0);
							}
				}
			L->sym_ok = pure;
			if ((rc = L->matsym.upload(fs))) return rc;
			{ // k_ps_sym's reciprocal eigenvalue sums: one table per distinct (plan, spacings) among the patches with pure axes
				std::map<std::tuple<int, double, double, double>, int> which;
				std::vector<int32_t>                                   itab(std::max(P, 1), 0);
				std::vector<double>                                    inv;
				for (int p = 0; p < P; p++) {
					const int k  = plan[p];
					bool      ok = true;
					for (int a = 0; a < 3; a++) ok &= (((keys[k] >> (2 * a)) & 1) == ((keys[k] >> (2 * a + 1)) & 1));
					if (!ok) continue;
					const auto key = std::make_tuple(k, rh2[(size_t) p * 3], rh2[(size_t) p * 3 + 1], rh2[(size_t) p * 3 + 2]);
					auto       it  = which.find(key);
					if (it == which.end()) {
						it = which.emplace(key, (int) which.size()).first;
						inv.resize(inv.size() + PSS_INV);
						double       *T  = &inv[(size_t) it->second * PSS_INV];
						const double *lx = &lam[((size_t) k * 3 + 0) * n], *ly = &lam[((size_t) k * 3 + 1) * n], *lz = &lam[((size_t) k * 3 + 2) * n];
						const double  rx = std::get<1>(key), ry = std::get<2>(key), rz = std::get<3>(key);
						for (int half = 0; half < 2; half++)
							for (int sl = 0; sl < 16; sl++)
								for (int pp = 0; pp < 2; pp++)
									for (int r = 0; r < 4; r++)
										for (int c = 0; c < 2; c++)
											for (int ln = 0; ln < 64; ln++) {
												const int    j = ln & 15, g = ln >> 4, kx = 2 * sl + half, ky = 2 * j + c, kz = 2 * (g + 4 * r) + pp;
												const double ex = lx[kx] * rx, ey = ly[ky] * ry, ez = lz[kz] * rz;
												const double d  = -((ex + ey) + ez); // (FftwPatchSolver.h:143-168: the eigenvalue of the patch operator)
												// zero mode of an all-Neumann patch: the coefficient is set to zero (FftwPatchSolver.h:197)
												T[((((size_t) (half * 16 + sl) * 2 + pp) * 4 + r) * 2 + c) * 64 + ln] = (zm[k] && kx == 0 && ky == 0 && kz == 0) ? 0.0 : 1.0 / d;
											}
					}
					itab[p] = it->second;
				}
				if (inv.empty()) inv.resize(1, 0.0);
				if ((rc = L->psinv.upload(inv)) || (rc = L->psitab.upload(itab))) return rc;
			}
			if (!pure) { // per-patch choice between k_ps_sym and k_ps_fused
				std::vector<int32_t> lst, mixed;
				for (int p = 0; p < P; p++) {
					bool ok = true;
					for (int a = 0; a < 3; a++) ok &= (((keys[plan[p]] >> (2 * a)) & 1) == ((keys[plan[p]] >> (2 * a + 1)) & 1));
					(ok ? lst : mixed).push_back(p);
				}
				L->n_pure = (int) lst.size();
				lst.insert(lst.end(), mixed.begin(), mixed.end());
				if ((rc = L->ps_list.upload(lst))) return rc;
			}
		}
		if (D == 2 && n <= 64) {
			std::vector<double> mt(mats.size());
			for (size_t m = 0; m < mats.size() / ((size_t) n * n); m++)
				for (int i = 0; i < n; i++)
					for (int j = 0; j < n; j++) mt[m * n * n + (size_t) j * n + i] = mats[m * n * n + (size_t) i * n + j];
			if ((rc = L->matsT.upload(mt))) return rc;
		}
		if (D == 2 && n == 64 && P > 0) { // k_patch_solve2d_sym's tables (see there): stage 0 / 1 forward x / y, 2 / 3 inverse x / y
			std::vector<double> fs((size_t) np * PS2S_PLAN, 0.0);
			std::vector<char>   pure(np, 1);
			for (int k = 0; k < np; k++) {
				for (int a = 0; a < 2; a++) pure[k] &= (((keys[k] >> (2 * a)) & 1) == ((keys[k] >> (2 * a + 1)) & 1));
				if (!pure[k]) continue;
				const double *Fx = &mats[((size_t) k * 4 + 0) * n * n], *Fy = &mats[((size_t) k * 4 + 1) * n * n];
				const double *Gx = &mats[((size_t) k * 4 + 2) * n * n], *Gy = &mats[((size_t) k * 4 + 3) * n * n];
				// the symmetry the kernel rests on: F[k][63 - j] = (-1)^k F[k][j], G[63 - j][k] = (-1)^k G[j][k]
				for (int i = 0; i < n && pure[k]; i++)
					for (int jj = 0; jj < n / 2; jj++) {
						const double sg = (i & 1) ? -1.0 : 1.0;
						const double e  = 1e-12;
						if (fabs(Fx[i * n + n - 1 - jj] - sg * Fx[i * n + jj]) > e || fabs(Fy[i * n + n - 1 - jj] - sg * Fy[i * n + jj]) > e
						    || fabs(Gx[(n - 1 - jj) * n + i] - sg * Gx[jj * n + i]) > e || fabs(Gy[(n - 1 - jj) * n + i] - sg * Gy[jj * n + i]) > e)
							pure[k] = 0;
					}
				if (!pure[k]) continue;
				double *S = &fs[(size_t) k * PS2S_PLAN];
				for (int ks = 0; ks < 8; ks++)
					for (int t = 0; t < 4; t++)
						for (int ln = 0; ln < 64; ln++) {
							const int    j = ln & 15, gq = ln >> 4, kk = 4 * ks + gq;
							const size_t e = ((size_t) ks * 4 + t) * 64 + ln;
							const int    wv = t < 2 ? 2 * (16 * t + j) : 2 * (16 * (t - 2) + j) + 1; // the wave number behind position 16 t + j
							S[0 * PS2S_STAGE + e] = Fx[wv * n + kk];
							S[1 * PS2S_STAGE + e] = Fy[wv * n + kk];
							S[2 * PS2S_STAGE + e] = Gx[(16 * (t & 1) + j) * n + 2 * kk + (t >> 1)];
							S[3 * PS2S_STAGE + e] = Gy[(16 * (t & 1) + j) * n + 2 * kk + (t >> 1)];
						}
			}
			{ // k_patch_solve2d_sym's reciprocal eigenvalues (times the transforms' scale), one table per distinct (plan, spacings), in the
			  // kernel's parity-split positions: position c < 32 holds wave number 2c, c >= 32 holds 2 (c - 32) + 1
				std::map<std::tuple<int, double, double>, int> which;
				std::vector<int32_t>                           itab(std::max(P, 1), 0);
				std::vector<double>                            inv;
				for (int p = 0; p < P; p++) {
					const int k = plan[p];
					if (!pure[k]) continue;
					const auto key = std::make_tuple(k, rh2[(size_t) p * 3], rh2[(size_t) p * 3 + 1]);
					auto       it  = which.find(key);
					if (it == which.end()) {
						it = which.emplace(key, (int) which.size()).first;
						inv.resize(inv.size() + (size_t) n * n);
						double       *T  = &inv[(size_t) it->second * n * n];
						const double *lx = &lam[((size_t) k * 2 + 0) * n], *ly = &lam[((size_t) k * 2 + 1) * n];
						const double  rx = std::get<1>(key), ry = std::get<2>(key), sc = 4.0 / ((double) n * n);
						for (int rp = 0; rp < n; rp++)
							for (int cp = 0; cp < n; cp++) {
								const int    ky = rp < 32 ? 2 * rp : 2 * (rp - 32) + 1, kx = cp < 32 ? 2 * cp : 2 * (cp - 32) + 1;
								const double d  = -(lx[kx] * rx + ly[ky] * ry);
								T[(size_t) rp * n + cp] = (zm[k] && kx == 0 && ky == 0) ? 0.0 : sc / d;
							}
					}
					itab[p] = it->second;
				}
				if (inv.empty()) inv.resize(1, 0.0);
				if ((rc = L->psinv.upload(inv)) || (rc = L->psitab.upload(itab))) return rc;
			}
			std::vector<int32_t> lst, mixed;
			for (int p = 0; p < P; p++) (pure[plan[p]] ? lst : mixed).push_back(p);
			L->n_pure2 = (int) lst.size();
			if (L->n_pure2 > 0 && (rc = L->mat2sym.upload(fs))) return rc;
			if (L->n_pure2 > 0 && L->n_pure2 < P) {
				lst.insert(lst.end(), mixed.begin(), mixed.end());
				if ((rc = L->ps2_list.upload(lst))) return rc;
			}
		}
		if ((rc = L->corr.alloc((size_t) std::max(P, 1) * NS * L->nf))) return rc;
		if ((rc = L->plan.upload(plan)) || (rc = L->mats.upload(mats)) || (rc = L->lam.upload(lam))
		    || (rc = L->zero_mode.upload(zm)))
			return rc;
	}

	// transfers to level li+1. A child (or a copy-through patch) whose parent lives on another rank
	// ships its restricted block there; the parent's rank ships octant blocks back for prolongation.
	// Canonical block order on both ends: (peer, parent patch (global), orthant).
	if (li + 1 < (int) H.levels.size()) {
		const Level         &cv = H.levels[li + 1];
		std::vector<int32_t> parent(P), orth(P), child((size_t) cv.P * NCH, -1), copy(cv.P, 0);
		struct Blk {
			int     peer, gpar, o, patch;
			int64_t size;
			bool    operator<(const Blk &b) const { return std::tie(peer, gpar, o) < std::tie(b.peer, b.gpar, b.o); }
		};
		std::vector<Blk> up, down;
		const bool       repl = cv.replicated && !lv.replicated;
		for (int p = 0; p < P; p++) {
			const int gp = lv.l2g[p], gpar = lv.g_parent[gp];
			orth[p]      = lv.g_orth_on_parent[gp];
			if (cv.g_rank[gpar] == me) {
				const int pc = cv.g_local[gpar];
				parent[p]    = pc;
				if (orth[p] < 0) {
					copy[pc]                  = 1;
					child[(size_t) pc * NCH] = p;
				} else {
					child[(size_t) pc * NCH + orth[p]] = p;
				}
			} else {
				up.push_back({cv.g_rank[gpar], gpar, orth[p] < 0 ? 0 : orth[p], p,
				              (int64_t) (orth[p] < 0 ? L->nc : L->nc / NCH)});
			}
		}
		for (int gf = 0; gf < lv.P_global; gf++) {
			const int gpar = lv.g_parent[gf];
			if (cv.g_rank[gpar] != me || lv.g_rank[gf] == me) continue;
			const int o = lv.g_orth_on_parent[gf];
			down.push_back({lv.g_rank[gf], gpar, o < 0 ? 0 : o, cv.g_local[gpar], (int64_t) (o < 0 ? L->nc : L->nc / NCH)});
			if (o < 0) copy[cv.g_local[gpar]] = 1;
		}
		std::sort(up.begin(), up.end());
		std::sort(down.begin(), down.end());
		std::vector<int32_t>                 upd, downd;
		std::vector<int64_t>                 upo, downo;
		std::vector<std::pair<int, int64_t>> ups, downs;
		int64_t                              pos = 0;
		for (size_t i = 0; i < up.size(); i++) {
			upd.push_back(up[i].patch);
			upd.push_back(orth[up[i].patch]);
			upo.push_back(pos);
			ups.emplace_back(up[i].peer, up[i].size);
			parent[up[i].patch] = -((int) i + 2); // prolong reads block i of upbuf
			pos += up[i].size;
		}
		std::vector<int32_t> bcd;
		if (repl) { // (up is empty: every parent is local) one block per local patch, in the order the receivers expect: (parent, orthant)
			std::vector<Blk> bc;
			for (int p = 0; p < P; p++)
				bc.push_back({0, lv.g_parent[lv.l2g[p]], orth[p] < 0 ? 0 : orth[p], p, (int64_t) (orth[p] < 0 ? L->nc : L->nc / NCH)});
			std::sort(bc.begin(), bc.end());
			for (size_t i = 0; i < bc.size(); i++) {
				upd.push_back(bc[i].patch); // (fine patch, orthant): k_restrict_pack restricts it into its block
				upd.push_back(orth[bc[i].patch]);
				bcd.push_back(parent[bc[i].patch]); // (coarse patch, orthant or -1): k_prolong_pack copies the finished octant out
				bcd.push_back(orth[bc[i].patch] < 0 ? -1 : bc[i].o);
				upo.push_back(pos);
				pos += bc[i].size;
			}
		}
		const int64_t up_total = pos;
		pos                    = 0;
		for (size_t i = 0; i < down.size(); i++) {
			const int pc = down[i].patch;
			const bool cp = down[i].size == (int64_t) L->nc;
			downd.push_back(pc);
			downd.push_back(cp ? -1 : down[i].o);
			downo.push_back(pos);
			downs.emplace_back(down[i].peer, down[i].size);
			child[(size_t) pc * NCH + (cp ? 0 : down[i].o)] = -((int) i + 2); // restrict reads block i of downbuf
			pos += down[i].size;
		}
		const int64_t down_total = pos;
		for (int pc = 0; pc < cv.P; pc++) {
			if (copy[pc]) continue;
			for (int o = 0; o < NCH; o++)
				if (child[(size_t) pc * NCH + o] == -1)
					return te::fail(TE_EINVAL, "te_gmg_create: coarse patch with a missing child");
		}
		L->Pc      = cv.P;
		// (repl: the blocks in `down` are received for the restriction only; every parent is local)
		const bool parents_local = up.empty() && (down.empty() || repl);
		L->prolong_fusable = (D == 3 && L->ncf == 0 && parents_local
		                      && std::all_of(orth.begin(), orth.end(), [](int32_t o) { return o >= 0; }));
		L->has_copy           = std::any_of(orth.begin(), orth.end(), [](int32_t o) { return o < 0; });
		L->prolong_fusable_cf = (D == 3 && parents_local && !g->cfg.has(O_NO_CFP));
		if (D == 2 && L->lds2d && parents_local) { // (no transfers, or a coarse level on every rank: its blocks travel behind the kernels)
			L->fuse2d          = true;
			// (faces on other ranks are fine: their values of u + P e arrive in ghost slots, packProlongFaces2d)
			L->prolong_fusable = ((g->cfg.has(O_2D_NO_MR_FUSE) ? L->nslots == 0 : L->ncf == 0)
			                      && std::all_of(orth.begin(), orth.end(), [](int32_t o) { return o >= 0; }));
		}
		if (D == 2 && L->lds2d) { // the 3D fusions in 2D (kernels2d.hpp)
			// a global fact, as in 3D (all ranks and every partition take the same arithmetic path): the level is uniformly
			// refined everywhere -- no coarse/fine face, every patch a quadrant child
			bool uniform = true;
			for (int gp = 0; gp < lv.P_global && uniform; gp++) {
				uniform = lv.g_orth_on_parent[gp] >= 0;
				for (int s2 = 0; s2 < NS && uniform; s2++) uniform = lv.g_nbr_kind[(size_t) gp * NS + s2] <= NBR_NORMAL;
			}
			L->fuse2_ok = uniform;
			if (uniform && (rc = L->e4buf.alloc((size_t) std::max(P, 1) * 4 * n))) return rc;
		}
		L->n_up    = (int) (upd.size() / 2);
		L->n_down  = (int) down.size();
		L->repl_up = repl;
		if (D == 3 && repl) {
			bool uniform = true;
			for (int gp = 0; gp < lv.P_global && uniform; gp++) {
				uniform = lv.g_orth_on_parent[gp] >= 0;
				for (int s2 = 0; s2 < NS && uniform; s2++) uniform = lv.g_nbr_kind[(size_t) gp * NS + s2] <= NBR_NORMAL;
			}
			L->post_exchange_free = uniform;
			if (uniform && nremote > 0) {
				std::vector<int32_t> sp(nremote), so(nremote);
				for (int i = 0; i < nremote; i++) {
					sp[i] = cv.g_local[lv.g_parent[recvs[i].nb]];
					so[i] = lv.g_orth_on_parent[recvs[i].nb];
				}
				if ((rc = L->slot_parent.upload(sp)) || (rc = L->slot_orth.upload(so))) return rc;
			}
			// in-place exchange of the restricted blocks: who fills which coarse patches
			std::vector<int> owner(cv.P_global, -1), lo(H.nranks, cv.P_global), hi(H.nranks, -1), cnt(H.nranks, 0);
			bool             direct = true;
			for (int gf = 0; gf < lv.P_global && direct; gf++) {
				int &o = owner[lv.g_parent[gf]];
				if (o >= 0 && o != lv.g_rank[gf]) direct = false;
				o = lv.g_rank[gf];
			}
			for (int pc = 0; pc < cv.P_global && direct; pc++) {
				const int r = owner[pc], lc = cv.g_local[pc];
				if (r < 0) {
					direct = false;
					break;
				}
				lo[r] = std::min(lo[r], lc), hi[r] = std::max(hi[r], lc), cnt[r]++;
			}
			for (int r = 0; r < H.nranks && direct; r++) direct = (cnt[r] == 0 || cnt[r] == hi[r] - lo[r] + 1);
			if (direct) {
				for (int r = 0; r < H.nranks; r++) {
					if (r == me || (cnt[r] == 0 && cnt[me] == 0)) continue;
					L->tx_direct.peers.push_back(r);
					L->tx_direct.send_off.push_back(cnt[me] ? (int64_t) lo[me] * (int64_t) L->nc : 0);
					L->tx_direct.send_cnt.push_back((int64_t) cnt[me] * (int64_t) L->nc);
					L->tx_direct.recv_off.push_back(cnt[r] ? (int64_t) lo[r] * (int64_t) L->nc : 0);
					L->tx_direct.recv_cnt.push_back((int64_t) cnt[r] * (int64_t) L->nc);
				}
				L->repl_direct = true;
			}
		}
		if (repl) {
			// restrict: the same range of upbuf to every other rank (if this rank has patches here at all), and from every rank
			// that has patches here its blocks; prolong: nothing
			L->tx_up = mergePlan({}, downs);
			ExPlan &pl = L->tx_up;
			if (up_total > 0) {
				ExPlan full;
				size_t k = 0;
				for (int r = 0; r < H.nranks; r++) {
					if (r == me) continue;
					while (k < pl.peers.size() && pl.peers[k] < r) k++;
					const bool have = k < pl.peers.size() && pl.peers[k] == r;
					full.peers.push_back(r);
					full.send_off.push_back(0);
					full.send_cnt.push_back(up_total);
					full.recv_off.push_back(have ? pl.recv_off[k] : 0);
					full.recv_cnt.push_back(have ? pl.recv_cnt[k] : 0);
				}
				pl = full;
			}
			L->tx_down = ExPlan();
			if ((rc = L->bc_desc.upload(bcd))) return rc;
		} else {
			L->tx_up   = mergePlan(ups, downs);   // restrict: send child blocks, receive into downbuf
			L->tx_down = mergePlan(downs, ups);   // prolong: send octants, receive into upbuf
		}
		if (L->prolong_fusable && D == 3) { // ProlongSrc::cbase: coarseOctant() of every patch and of its six neighbours, precomputed
			const int64_t nn = (int64_t) n * n, nnn = nn * n, hh = n / 2;
			auto          base = [&](int p) {
                const int o = orth[p];
                return (int64_t) parent[p] * nnn + ((o & 1) ? hh : 0) + n * ((o & 2) ? hh : 0) + nn * ((o & 4) ? hh : 0);
			};
			std::vector<int64_t> cb((size_t) std::max(P, 1) * 7, -1);
			for (int p = 0; p < P; p++) {
				cb[(size_t) p * 7] = base(p);
				for (int s2 = 0; s2 < 6; s2++)
					if (fk[(size_t) p * 6 + s2] == FACE_LOCAL) cb[(size_t) p * 7 + 1 + s2] = base(fs[(size_t) p * 6 + s2]);
			}
			if ((rc = L->cbase.upload(cb))) return rc;
		}
		if ((rc = L->parent.upload(parent)) || (rc = L->orth.upload(orth)) || (rc = L->child.upload(child))
		    || (rc = L->copy.upload(copy)) || (rc = L->up_desc.upload(upd)) || (rc = L->down_desc.upload(downd))
		    || (rc = L->up_off.upload(upo)) || (rc = L->down_off.upload(downo))
		    || (rc = L->upbuf.alloc((size_t) std::max<int64_t>(up_total, 1)))
		    || (rc = L->downbuf.alloc((size_t) std::max<int64_t>(down_total, 1))))
			return rc;
	}
	g->levels.push_back(std::move(L));
	return TE_OK;
}

int newVec(te_gmg *g, int level, te_vec **out)
{
	LevelHost &L = *g->levels[level];
	auto       v = new te_vec;
	v->g         = g;
	v->level     = level;
	v->n         = (size_t) L.P * L.nc;
	hipError_t e = hipMalloc(&v->d, sizeof(double) * std::max<size_t>(v->n, 2));
	if (e != hipSuccess) {
		delete v;
		return te::fail(TE_EHIP, std::string("hipMalloc(vector): ") + hipGetErrorString(e));
	}
	e = hipMemsetAsync(v->d, 0, sizeof(double) * v->n, g->stream);
	if (e != hipSuccess) {
		(void) hipFree(v->d);
		delete v;
		return te::fail(TE_EHIP, std::string("hipMemsetAsync: ") + hipGetErrorString(e));
	}
	*out = v;
	return TE_OK;
}

} // namespace tei

extern "C" {
void te_cycle_opts_default(te_cycle_opts *o)
{
	if (!o) return;
	o->pre_sweeps = o->post_sweeps = o->coarse_sweeps = o->mid_sweeps = 1; // CycleOpts.h:64-79
	o->cycle_type   = 0;
	o->smoother     = TE_SMOOTH_PATCH_SOLVE;
	o->omega        = 6.0 / 7.0;
	o->exact_coarse = 1;
	o->fuse         = 3;
}

int te_gmg_create(const te_hier *h, int device, te_gmg **out)
{
	return guarded([&]() -> int {
		if (!h || !out) return te::fail(TE_EINVAL, "te_gmg_create: null argument");
		using clk = std::chrono::steady_clock;
		auto ms   = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
		const auto t_begin = clk::now();
		SetupAcc   acc;
		struct AccScope { // (the allocations and uploads of THIS call, whichever way it ends)
			explicit AccScope(SetupAcc *a) { g_setup_acc = a; }
			~AccScope() { g_setup_acc = nullptr; }
		} acc_scope(&acc);
		int ndev = 0;
		if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
			return te::fail(TE_EHIP, "te_gmg_create: no HIP device visible (this library has no CPU fallback)");
		if (device < 0) HIPCHK(hipGetDevice(&device));
		HIPCHK(hipSetDevice(device));
		auto g    = std::make_unique<te_gmg>();
		g->device = device;
		g->dim    = h->h.dim;
		g->n      = h->h.n;
		g->rank   = h->h.rank;
		g->nranks = h->h.nranks;
		g->placement[0] = h->h.agglomerate, g->placement[1] = h->h.agglomerate_max, g->placement[2] = h->h.replicate;
		g->placement[3] = (double) h->h.levels.size();
		memset(g->calls, 0, sizeof(g->calls));
		memset(g->cells, 0, sizeof(g->cells));
		memset(g->total_ms, 0, sizeof(g->total_ms));
		HIPCHK(hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
		HIPCHK(hipStreamCreateWithFlags(&g->comm_stream, hipStreamNonBlocking));
		HIPCHK(hipEventCreateWithFlags(&g->ev_pack, hipEventDisableTiming));
		HIPCHK(hipEventCreateWithFlags(&g->ev_recv, hipEventDisableTiming));
		g->cfg.fromEnv();
		g->overlap = !g->cfg.has(O_NO_OVERLAP);
		const auto t_ctx = clk::now();
		int rc;
		for (int li = 0; li < (int) h->h.levels.size(); li++)
			if ((rc = buildLevel(g.get(), h->h, li))) return rc;
		const auto     t_levels = clk::now();
		const SetupAcc acc_levels = acc;
		{ // partial sums: the reduction kernels' blocks, or one pair per work item of a stencil launch with fused sums (<= 8 slabs per patch)
			size_t items = (size_t) g->red_blocks;
			for (auto &L : g->levels) items = std::max(items, (size_t) L->P * (L->P <= 64 ? 8 : (L->P < 2048 ? 4 : 1)));
			// 2D: one pair per workgroup of k_stencil2d's FLAT grid (one x-pair per thread: a capped grid-stride loop streams a third
			// slower on this chip and cost the fused sums more than the passes they replace)
			for (auto &L : g->levels)
				if (L->dim == 2) items = std::max(items, ((size_t) L->P * L->nc / 2 + 255) / 256);
			if ((rc = g->partial.alloc(2 * items)) || (rc = g->result.alloc(8))) return rc;
		}
		HIPCHK(hipHostMalloc((void **) &g->result_host, 8 * sizeof(double), hipHostMallocDefault));
		for (int li = 0; li < (int) g->levels.size(); li++) {
			LevelHost &L = *g->levels[li];
			te_vec    *v;
			if ((rc = newVec(g.get(), li, &v))) return rc;
			L.r.reset(v);
			if ((rc = newVec(g.get(), li, &v))) return rc;
			L.t.reset(v);
			if (li > 0) {
				if ((rc = newVec(g.get(), li, &v))) return rc;
				L.u.reset(v);
				if ((rc = newVec(g.get(), li, &v))) return rc;
				L.f.reset(v);
			}
		}
		const auto t_vecs = clk::now();
		HIPCHK(hipStreamSynchronize(g->stream));
		const auto t_end = clk::now();
		g->setup_ms[0]   = ms(t_begin, t_ctx);
		g->setup_ms[1]   = ms(t_ctx, t_levels) - acc_levels.malloc_ms - acc_levels.copy_ms;
		g->setup_ms[2]   = acc.malloc_ms;
		g->setup_ms[3]   = acc.nmalloc;
		g->setup_ms[4]   = acc.copy_ms;
		g->setup_ms[5]   = ms(t_levels, t_vecs) - (acc.malloc_ms - acc_levels.malloc_ms) - (acc.copy_ms - acc_levels.copy_ms);
		g->setup_ms[6]   = ms(t_vecs, t_end);
		g->setup_ms[7]   = ms(t_begin, t_end);
		*out = g.release();
		return TE_OK;
	});
}

int te_gmg_setup_ms(const te_gmg *g, double *out, int n)
{
	return guarded([&]() -> int {
		if (!g || !out || n < 0) return te::fail(TE_EINVAL, "te_gmg_setup_ms: bad argument");
		for (int i = 0; i < std::min(n, 8); i++) out[i] = g->setup_ms[i];
		return TE_OK;
	});
}

void te_gmg_destroy(te_gmg *g)
{
	if (!g) return;
	watchdogStop(g);
	(void) hipStreamSynchronize(g->stream);
	if (g->comm_stream) (void) hipStreamSynchronize(g->comm_stream);
	pushTeardown(g, true); // (the coarse vectors get their own storage back before they are freed below)
	for (auto &L : g->levels) {
		for (te_vec *v : {L->u.get(), L->f.get(), L->r.get(), L->t.get()})
			if (v && v->d) (void) hipFree(v->d);
	}
	for (te_vec *v : g->bicg_work)
		if (v) te_vec_destroy(v);
	for (auto &e : g->ev_pool) {
		(void) hipEventDestroy(e.a);
		(void) hipEventDestroy(e.b);
	}
	if (g->rccl.comm && g->rccl.CommDestroy) (void) g->rccl.CommDestroy(g->rccl.comm);
	if (g->result_host) (void) hipHostFree(g->result_host);
	if (g->ev_pack) (void) hipEventDestroy(g->ev_pack);
	if (g->ev_recv) (void) hipEventDestroy(g->ev_recv);
	if (g->comm_stream) (void) hipStreamDestroy(g->comm_stream);
	(void) hipStreamDestroy(g->stream);
	delete g;
}

int   te_gmg_num_levels(const te_gmg *g) { return guarded([&]() -> int { return g ? (int) g->levels.size() : TE_EINVAL; }); }

int   te_gmg_sync(te_gmg *g)
{
	return guarded([&]() -> int {
		if (!g) return te::fail(TE_EINVAL, "te_gmg_sync: null");
		HIPCHK(hipStreamSynchronize(g->stream));
		return TE_OK;
	});
}

void *te_gmg_stream(te_gmg *g) { return g ? (void *) g->stream : nullptr; }

int te_vec_create(te_gmg *g, int level, te_vec **out)
{
	return guarded([&]() -> int {
		if (!g || !out || level < 0 || level >= (int) g->levels.size())
			return te::fail(TE_EINVAL, "te_vec_create: bad argument");
		HIPCHK(hipSetDevice(g->device));
		return newVec(g, level, out);
	});
}

void te_vec_destroy(te_vec *v)
{
	if (!v) return;
	(void) hipStreamSynchronize(v->g->stream);
	(void) hipFree(v->d);
	delete v;
}

size_t te_vec_size(const te_vec *v) { return v ? v->n : 0; }

int    te_vec_upload(te_vec *v, const double *host)
{
	return guarded([&]() -> int {
		if (!v || !host) return te::fail(TE_EINVAL, "te_vec_upload: null");
		HIPCHK(hipMemcpyAsync(v->d, host, sizeof(double) * v->n, hipMemcpyHostToDevice, v->g->stream));
		HIPCHK(hipStreamSynchronize(v->g->stream));
		return TE_OK;
	});
}

int te_vec_download(const te_vec *v, double *host)
{
	return guarded([&]() -> int {
		if (!v || !host) return te::fail(TE_EINVAL, "te_vec_download: null");
		HIPCHK(hipMemcpyAsync(host, v->d, sizeof(double) * v->n, hipMemcpyDeviceToHost, v->g->stream));
		HIPCHK(hipStreamSynchronize(v->g->stream));
		return TE_OK;
	});
}

void *te_vec_device_ptr(te_vec *v) { return v ? v->d : nullptr; }

// Init::initDirichlet / initNeumann for the drivers' canned problems, on the device (initkernels.hpp)
int te_init_problem(te_gmg *g, int level, int problem, int neumann, te_vec *f, te_vec *exact)
{
	return guarded([&]() -> int {
		int rc;
		if ((rc = checkLevelVec(g, level, f, "te_init_problem"))) return rc;
		if (exact && (rc = checkLevelVec(g, level, exact, "te_init_problem"))) return rc;
		if (exact == f) return te::fail(TE_EINVAL, "te_init_problem: f and exact must be different vectors");
		LevelHost &L = *g->levels[level];
		if (L.xf_valid_for == f->d || (exact && L.xf_valid_for == exact->d)) L.xf_valid_for = nullptr;
		if (L.P == 0) return TE_OK;
		InitGeom G;
		G.dim = L.dim, G.n = L.n, G.P = L.P;
		G.starts = L.geom_starts.p, G.h = L.geom_h.p, G.face_kind = L.face_kind.p, G.ids = L.node_ids.p;
		const dim3 grid(gridFor(f->n, 256, 1 << 20)), blk(256);
		double    *e = exact ? exact->d : nullptr;
		Timed      t(g, KC_VECOP, f->n);
#define TE_INIT(K, PROB)                                                                              \
		if (neumann)                                                                                      \
			hipLaunchKernelGGL((K<PROB, true>), grid, blk, 0, g->stream, G, f->d, e);                     \
		else                                                                                              \
			hipLaunchKernelGGL((K<PROB, false>), grid, blk, 0, g->stream, G, f->d, e);
		if (problem == PROBLEM_RANDOM) {
			hipLaunchKernelGGL(k_init_random, grid, blk, 0, g->stream, G, L.nc, (uint64_t) 0x5EED, f->d, e);
		} else if (problem == PROBLEM_TRIG) {
			if (L.dim == 3) {
				TE_INIT(k_init3d, PROBLEM_TRIG)
			} else {
				TE_INIT(k_init2d, PROBLEM_TRIG)
			}
		} else if (problem == PROBLEM_GAUSS) {
			if (L.dim == 3) {
				TE_INIT(k_init3d, PROBLEM_GAUSS)
			} else {
				TE_INIT(k_init2d, PROBLEM_GAUSS)
			}
		} else {
			return te::fail(TE_EINVAL, "te_init_problem: unknown problem");
		}
#undef TE_INIT
		HIPCHK(hipGetLastError());
		return TE_OK;
	});
}

// Vector<D>::getLocalData(i) for a run of patches (PetscVector.h:87-98): what Init::initDirichlet, the writers and
// the C++ adaptor's host mirror move -- never the whole vector for one patch.
int te_vec_upload_patches(te_vec *v, int first_patch, int npatches, const double *host)
{
	return guarded([&]() -> int {
		if (!v || !host) return te::fail(TE_EINVAL, "te_vec_upload_patches: null");
		const size_t nc = v->g->levels[v->level]->nc;
		if (first_patch < 0 || npatches < 0 || ((size_t) first_patch + npatches) * nc > v->n)
			return te::fail(TE_EINVAL, "te_vec_upload_patches: patch range outside the vector");
		if (npatches == 0) return TE_OK;
		LevelHost &L = *v->g->levels[v->level];
		if (L.xf_valid_for == v->d) L.xf_valid_for = nullptr;
		HIPCHK(hipMemcpyAsync(v->d + (size_t) first_patch * nc, host, sizeof(double) * nc * npatches, hipMemcpyHostToDevice, v->g->stream));
		HIPCHK(hipStreamSynchronize(v->g->stream));
		return TE_OK;
	});
}

int te_vec_download_patches(const te_vec *v, int first_patch, int npatches, double *host)
{
	return guarded([&]() -> int {
		if (!v || !host) return te::fail(TE_EINVAL, "te_vec_download_patches: null");
		const size_t nc = v->g->levels[v->level]->nc;
		if (first_patch < 0 || npatches < 0 || ((size_t) first_patch + npatches) * nc > v->n)
			return te::fail(TE_EINVAL, "te_vec_download_patches: patch range outside the vector");
		if (npatches == 0) return TE_OK;
		HIPCHK(hipMemcpyAsync(host, v->d + (size_t) first_patch * nc, sizeof(double) * nc * npatches, hipMemcpyDeviceToHost, v->g->stream));
		HIPCHK(hipStreamSynchronize(v->g->stream));
		return TE_OK;
	});
}

int te_gmg_profile(te_gmg *g, int enable)
{
	return guarded([&]() -> int {
		if (!g) return te::fail(TE_EINVAL, "te_gmg_profile: null");
		drainEvents(g);
		g->profiling = enable != 0;
		return TE_OK;
	});
}

int te_integrate(te_gmg *g, int level, const te_vec *v, double *out)
{
	return guarded([&]() -> int {
		int rc;
		if (!out) return te::fail(TE_EINVAL, "te_integrate: null result");
		if ((rc = checkLevelVec(g, level, v, "te_integrate"))) return rc;
		LevelHost &L = *g->levels[level];
		*out         = 0.0;
		if (L.P == 0 || (L.replicated && g->rank != 0)) return TE_OK; // (a level on every rank counts once: rank 0's)
		DevBuf<double> part;
		if ((rc = part.alloc(L.P))) return rc;
		hipLaunchKernelGGL(k_patch_integrals, dim3(L.P), dim3(256), 0, g->stream, (int) L.nc, v->d, L.cellvol.p, part.p);
		HIPCHK(hipGetLastError());
		std::vector<double> h(L.P);
		HIPCHK(hipMemcpyAsync(h.data(), part.p, sizeof(double) * L.P, hipMemcpyDeviceToHost, g->stream));
		HIPCHK(hipStreamSynchronize(g->stream));
		double sum = 0.0;
		for (double x : h) sum += x; // patch order, as the reference's loop over its patch map
		*out = sum;
		return TE_OK;
	});
}

int te_volume(te_gmg *g, int level, double *out)
{
	return guarded([&]() -> int {
		if (!g || !out || level < 0 || level >= (int) g->levels.size()) return te::fail(TE_EINVAL, "te_volume: bad argument");
		double sum = 0.0;
		if (!(g->levels[level]->replicated && g->rank != 0)) // (a level on every rank counts once: rank 0's)
			for (double x : g->levels[level]->patch_vol) sum += x;
		*out = sum;
		return TE_OK;
	});
}

int te_gmg_profile_select(te_gmg *g, const char *name)
{
	return guarded([&]() -> int {
		if (!g) return te::fail(TE_EINVAL, "te_gmg_profile_select: null");
		drainEvents(g);
		g->prof_only = -1;
		if (!name || !*name) return TE_OK;
		for (int k = 0; k < KC_COUNT; k++)
			if (!strcmp(name, kclassName[k])) {
				g->prof_only = k;
				return TE_OK;
			}
		return te::fail(TE_EINVAL, std::string("te_gmg_profile_select: unknown kernel class ") + name);
	});
}

int te_gmg_profile_stride(te_gmg *g, int stride)
{
	return guarded([&]() -> int {
		if (!g) return te::fail(TE_EINVAL, "te_gmg_profile_stride: null");
		drainEvents(g);
		g->prof_stride = stride > 1 ? stride : 1;
		memset(g->prof_seq, 0, sizeof(g->prof_seq));
		return TE_OK;
	});
}

int te_gmg_profile_reset(te_gmg *g)
{
	return guarded([&]() -> int {
		if (!g) return te::fail(TE_EINVAL, "te_gmg_profile_reset: null");
		drainEvents(g);
		memset(g->calls, 0, sizeof(g->calls));
		memset(g->cells, 0, sizeof(g->cells));
		memset(g->total_ms, 0, sizeof(g->total_ms));
		return TE_OK;
	});
}

int te_gmg_profile_rows(te_gmg *g, int max_rows, char (*name)[64], int64_t *calls, double *total_ms,
                        int64_t *cells)
{
	return guarded([&]() -> int {
		if (!g || !name || !calls || !total_ms || !cells) return te::fail(TE_EINVAL, "te_gmg_profile_rows: null");
		drainEvents(g);
		int n = 0;
		for (int k = 0; k < KC_COUNT && n < max_rows; k++) {
			if (g->calls[k] == 0) continue;
			strncpy(name[n], kclassName[k], 63);
			name[n][63] = 0;
			calls[n]    = g->calls[k];
			total_ms[n] = g->total_ms[k];
			cells[n]    = g->cells[k];
			n++;
		}
		return n;
	});
}

#if TE_STAMPS
// diagnostic build only (not in include/te_hip.h): te_stamps_begin clears and arms the collection; te_stamps_read returns the number of
// instrumented launches since then and copies their stamps ([launch][1024][8] ticks of 10 ns), names and workgroup counts
int te_stamps_begin(te_gmg *g)
{
	return guarded([&]() -> int {
		auto &S = g->stamps;
		int   rc;
		if (!S.buf.p && (rc = S.buf.alloc((size_t) S.MAXL * S.MAXWG * TE_NSTAMP))) return rc;
		HIPCHK(hipStreamSynchronize(g->stream));
		HIPCHK(hipMemset(S.buf.p, 0, sizeof(unsigned long long) * S.buf.n));
		S.names.clear(), S.wgs.clear();
		S.on = true;
		return TE_OK;
	});
}
int te_stamps_read(te_gmg *g, unsigned long long *out, char (*names)[64], int *wgs, int max_launches)
{
	return guarded([&]() -> int {
		auto &S = g->stamps;
		S.on    = false;
		HIPCHK(hipStreamSynchronize(g->stream));
		const int n = std::min((int) S.names.size(), max_launches);
		if (n > 0) HIPCHK(hipMemcpy(out, S.buf.p, sizeof(unsigned long long) * (size_t) n * S.MAXWG * TE_NSTAMP, hipMemcpyDeviceToHost));
		for (int i = 0; i < n; i++) {
			strncpy(names[i], S.names[i].c_str(), 63);
			names[i][63] = 0;
			wgs[i]       = S.wgs[i];
		}
		return n;
	});
}
#endif

// te_bicgstab keeps its eight level-0 work vectors between solves (8 GiB at 512^3); a caller that is done solving hands
// them back with this call (they are allocated again by the next te_bicgstab)
int te_gmg_release_workspace(te_gmg *g)
{
	return guarded([&]() -> int {
			if (!g) return te::fail(TE_EINVAL, "te_gmg_release_workspace: null");
			for (te_vec *&v : g->bicg_work) {
				if (v) te_vec_destroy(v);
				v = nullptr;
			}
			return TE_OK;
	});
}

// One TE_* switch (docs/SWITCHES.md) of this solver: value == NULL clears it (back to the default). te_gmg_create reads all of
// them from the environment once; afterwards this is the only way to change one. Switches that shape the level tables
// (TE_2D_SIMPLE, TE_NO_CFP, TE_2D_NO_MR_FUSE, TE_NO_OVERLAP, TE_EXCHANGE_TIMEOUT) are fixed at creation: TE_ESTATE.
int te_gmg_set_option(te_gmg *g, const char *name, const char *value)
{
	return guarded([&]() -> int {
			if (!g || !name) return te::fail(TE_EINVAL, "te_gmg_set_option: null argument");
			for (int o = 0; o < O_COUNT; o++)
				if (!strcmp(name, optName[o])) {
					if (optStructural(o))
						return te::fail(TE_ESTATE, std::string("te_gmg_set_option: ") + name + " is read when the solver is created; set it in the environment before te_gmg_create");
					g->cfg.set(o, value);
					if (o == O_PUSH_TIMEOUT) g->push.timeout_s = value ? std::max(0.1, atof(value)) : (g->wd.timeout_s > 0 ? g->wd.timeout_s : 300.0);
					g->verified_opts.clear(); // (an option may change which exchanges a cycle issues)
					return TE_OK;
				}
			return te::fail(TE_EINVAL, std::string("te_gmg_set_option: unknown option ") + name);
	});
}
} // extern "C"
