// TEST INFRASTRUCTURE — NOT PRODUCT CODE. See te_oracle.h for scope, citations and the
// pinning status of every function. Plain scalar C++ (OpenMP over patches only), written to
// follow the reference's arithmetic order, not to be fast.
#include "te_oracle.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace
{
int g_threads = 1;

inline int ipow(int b, int e)
{
	int r = 1;
	while (e-- > 0) r *= b;
	return r;
}
struct Geo {
	int dim, n, nsides, nc, nf; // cells per patch, cells per face
	int stride[3];
	explicit Geo(const orc_level *L)
	{
		dim       = L->dim;
		n         = L->n;
		nsides    = 2 * dim;
		nc        = ipow(n, dim);
		nf        = ipow(n, dim - 1);
		stride[0] = 1;
		stride[1] = n;
		stride[2] = n * n;
	}
	// offset of face cell (a,b) of side s, `off` layers in from the face
	// (Vector.h:152-177 getSliceOnSidePriv)
	inline int faceCell(int s, int a, int b, int off) const
	{
		int ax  = s / 2;
		int pos = (s & 1) ? (n - 1 - off) : off;
		int fa[2] = {0, 0}, k = 0;
		for (int i = 0; i < dim; i++)
			if (i != ax) fa[k++] = i;
		int idx = pos * stride[ax] + a * stride[fa[0]];
		if (dim == 3) idx += b * stride[fa[1]];
		return idx;
	}
};

// Interface numbering: SchurInfo.h:141-150 (normal), :229-237 (coarse), :322-331 (fine);
// dense local indices in first-seen order as SchurHelper.h:377-397 assigns them.
struct Ifaces {
	std::vector<int> own;    // [P*nsides] index of the interface on (p,s) or -1
	std::vector<int> other;  // [P*nsides*4] coarse: [0] = coarse patch's iface; fine: 4 fine ifaces
	int              count = 0;
};
Ifaces buildIfaces(const orc_level *L)
{
	Geo                G(L);
	Ifaces             I;
	std::map<int, int> rev;
	auto               get = [&](int id) {
        auto it = rev.find(id);
        if (it != rev.end()) return it->second;
        int v   = I.count++;
        rev[id] = v;
        return v;
	};
	const int NS = G.nsides;
	I.own.assign((size_t) L->P * NS, -1);
	I.other.assign((size_t) L->P * NS * 4, -1);
	for (int p = 0; p < L->P; p++) {
		for (int s = 0; s < NS; s++) {
			size_t f    = (size_t) p * NS + s;
			int    kind = L->nbr_kind[f];
			if (kind == 1) {
				int id = (s & 1) ? L->id[L->nbr[f * 4]] * NS + (s ^ 1) : L->id[p] * NS + s;
				I.own[f] = get(id);
			} else if (kind == 2) {
				I.own[f]       = get(L->id[p] * NS + s);
				I.other[f * 4] = get(L->id[L->nbr[f * 4]] * NS + (s ^ 1));
			} else if (kind == 3) {
				I.own[f] = get(L->id[p] * NS + s);
				for (int q = 0; q < (1 << (G.dim - 1)); q++)
					I.other[f * 4 + q] = get(L->id[L->nbr[f * 4 + q]] * NS + (s ^ 1));
			}
		}
	}
	return I;
}

// ---- a6: one patch's contributions -----------------------------------------------------
void interpPatch(const orc_level *L, const Geo &G, const Ifaces &I, int p, const double *u,
                 double *gamma)
{
	const int     n  = G.n;
	const double *up = u + (size_t) p * G.nc;
	for (int s = 0; s < G.nsides; s++) {
		size_t f    = (size_t) p * G.nsides + s;
		int    kind = L->nbr_kind[f];
		if (kind == 0) continue;
		auto sl = [&](int a, int b) { return up[G.faceCell(s, a, b, 0)]; };
		if (G.dim == 3) {
			auto at = [&](int iface, int a, int b) -> double & {
				return gamma[(size_t) iface * G.nf + a + n * b];
			};
			if (kind == 1) { // normal, TriLinInterp.cpp:78-84
				for (int b = 0; b < n; b++)
					for (int a = 0; a < n; a++) at(I.own[f], a, b) += 0.5 * sl(a, b);
			} else if (kind == 2) {
				// fine_to_fine, TriLinInterp.cpp:85-98
				for (int b = 0; b < n / 2; b++)
					for (int a = 0; a < n / 2; a++) {
						double va = sl(2 * a, 2 * b), vb = sl(2 * a + 1, 2 * b);
						double vc = sl(2 * a, 2 * b + 1), vd = sl(2 * a + 1, 2 * b + 1);
						at(I.own[f], 2 * a, 2 * b) += (11 * va - vb - vc - vd) / 12.0;
						at(I.own[f], 2 * a + 1, 2 * b) += (-va + 11 * vb - vc - vd) / 12.0;
						at(I.own[f], 2 * a, 2 * b + 1) += (-va - vb + 11 * vc - vd) / 12.0;
						at(I.own[f], 2 * a + 1, 2 * b + 1) += (-va - vb - vc + 11 * vd) / 12.0;
					}
				// fine_to_coarse, TriLinInterp.cpp:138-170
				int q = L->nbr_orth[f], oa = (q & 1) ? n : 0, ob = (q & 2) ? n : 0;
				for (int b = 0; b < n; b++)
					for (int a = 0; a < n; a++)
						at(I.other[f * 4], (a + oa) / 2, (b + ob) / 2) += 1.0 / 6.0 * sl(a, b);
			} else {
				// coarse_to_coarse, TriLinInterp.cpp:131-137
				for (int b = 0; b < n; b++)
					for (int a = 0; a < n; a++) at(I.own[f], a, b) += 2.0 / 6.0 * sl(a, b);
				// coarse_to_fine, TriLinInterp.cpp:99-130
				for (int q = 0; q < 4; q++) {
					int oa = (q & 1) ? n : 0, ob = (q & 2) ? n : 0;
					for (int b = 0; b < n; b++)
						for (int a = 0; a < n; a++)
							at(I.other[f * 4 + q], a, b) += 4.0 * sl((a + oa) / 2, (b + ob) / 2) / 12.0;
				}
			}
		} else {
			auto at = [&](int iface, int a) -> double & { return gamma[(size_t) iface * G.nf + a]; };
			if (kind == 1) { // BilinearInterpolator.cpp:71-75
				for (int a = 0; a < n; a++) at(I.own[f], a) += 0.5 * sl(a, 0);
			} else if (kind == 2) {
				// fine_to_fine :95-103
				for (int a = 0; a < n; a += 2) at(I.own[f], a) += 5.0 / 6 * sl(a, 0) - 1.0 / 6 * sl(a + 1, 0);
				for (int a = 1; a < n; a += 2) at(I.own[f], a) += 5.0 / 6 * sl(a, 0) - 1.0 / 6 * sl(a - 1, 0);
				// fine_to_coarse :82-94
				int oa = (L->nbr_orth[f] & 1) ? n : 0;
				for (int a = 0; a < n; a += 2)
					at(I.other[f * 4], (oa + a) / 2) += 1.0 / 3 * sl(a, 0) + 1.0 / 3 * sl(a + 1, 0);
			} else {
				for (int a = 0; a < n; a++) at(I.own[f], a) += 1.0 / 3 * sl(a, 0); // :76-81
				for (int q = 0; q < 2; q++) {                                       // :104-115
					int oa = q ? n : 0;
					for (int a = 0; a < n; a++) at(I.other[f * 4 + q], a) += 2.0 / 6 * sl((oa + a) / 2, 0);
				}
			}
		}
	}
}

// ---- a3 / a4: one patch ----------------------------------------------------------------
// with_gamma = false reproduces StarPatchOp::apply (neighbour faces treated as Dirichlet).
void applyPatch(const orc_level *L, const Geo &G, const Ifaces *I, int p, const double *u,
                const double *gamma, double *f, bool with_gamma)
{
	const int     n  = G.n;
	const double *up = u + (size_t) p * G.nc;
	double       *fp = f + (size_t) p * G.nc;
	const int     nz = (G.dim == 3) ? n : 1;
	for (int ax = 0; ax < G.dim; ax++) {
		double    h2 = L->h[(size_t) p * G.dim + ax];
		h2 *= h2;
		const int st = G.stride[ax];
		const int sl = 2 * ax, su = 2 * ax + 1;
		const int kl = L->nbr_kind[(size_t) p * G.nsides + sl], ku = L->nbr_kind[(size_t) p * G.nsides + su];
		const bool nl = (L->neumann[p] >> sl) & 1, nu = (L->neumann[p] >> su) & 1;
		for (int z = 0; z < nz; z++)
			for (int y = 0; y < n; y++)
				for (int x = 0; x < n; x++) {
					int c[3] = {x, y, z};
					int idx  = x + n * y + n * n * z;
					// face coordinates = remaining axes in order
					int a = 0, b = 0;
					{
						int k = 0, fa[2] = {0, 0};
						for (int i = 0; i < G.dim; i++)
							if (i != ax) fa[k++] = c[i];
						a = fa[0];
						b = (G.dim == 3) ? fa[1] : 0;
					}
					double val;
					double mid = up[idx];
					if (c[ax] == 0) {
						double upper = up[idx + st];
						if (with_gamma && kl != 0) {
							double bnd = gamma[(size_t) I->own[(size_t) p * G.nsides + sl] * G.nf + a + n * b];
							val        = (2 * bnd - 3 * mid + upper) / h2;
						} else if (nl && (kl == 0 || !with_gamma)) {
							val = (-mid + upper) / h2;
						} else {
							val = (-3 * mid + upper) / h2;
						}
					} else if (c[ax] == n - 1) {
						double lower = up[idx - st];
						if (with_gamma && ku != 0) {
							double bnd = gamma[(size_t) I->own[(size_t) p * G.nsides + su] * G.nf + a + n * b];
							val        = (lower - 3 * mid + 2 * bnd) / h2;
						} else if (nu && (ku == 0 || !with_gamma)) {
							val = (lower - mid) / h2;
						} else {
							val = (lower - 3 * mid) / h2;
						}
					} else {
						double lower = up[idx - st], upper = up[idx + st];
						val = (lower - 2 * mid + upper) / h2;
					}
					if (ax == 0)
						fp[idx] = val;
					else
						fp[idx] += val;
				}
	}
}

// ---- a9: dense DST/DCT patch solve -------------------------------------------------------
enum DftType { DCT_II, DCT_III, DCT_IV, DST_II, DST_III, DST_IV };
// DftPatchSolver.h:237-289; returned row-major M with y_i = sum_j M[i*n+j] x_j
std::vector<double> transformMatrix(DftType t, int n)
{
	std::vector<double> m((size_t) n * n, 0.0);
	switch (t) {
		case DCT_II:
			for (int j = 0; j < n; j++)
				for (int i = 0; i < n; i++) m[i * n + j] = cos(M_PI / n * (i * (j + 0.5)));
			break;
		case DCT_III:
			for (int i = 0; i < n; i++) m[i * n] = 0.5;
			for (int j = 1; j < n; j++)
				for (int i = 0; i < n; i++) m[i * n + j] = cos(M_PI / n * ((i + 0.5) * j));
			break;
		case DCT_IV:
			for (int j = 0; j < n; j++)
				for (int i = 0; i < n; i++) m[i * n + j] = cos(M_PI / n * ((i + 0.5) * (j + 0.5)));
			break;
		case DST_II:
			for (int i = 0; i < n; i++)
				for (int j = 0; j < n; j++) m[i * n + j] = sin(M_PI / n * ((i + 1) * (j + 0.5)));
			break;
		case DST_III:
			for (int i = 0; i < n; i += 2) m[i * n + n - 1] = 0.5;
			for (int i = 1; i < n; i += 2) m[i * n + n - 1] = -0.5;
			for (int i = 0; i < n; i++)
				for (int j = 0; j < n - 1; j++) m[i * n + j] = sin(M_PI / n * ((i + 0.5) * (j + 1)));
			break;
		case DST_IV:
			for (int j = 0; j < n; j++)
				for (int i = 0; i < n; i++) m[i * n + j] = sin(M_PI / n * ((i + 0.5) * (j + 0.5)));
			break;
	}
	return m;
}
// out = transform of `in` along axis ax (DftPatchSolver.h:295-347: one dgemv per line)
void transformAxis(const Geo &G, const std::vector<double> &M, int ax, const double *in, double *out)
{
	const int n = G.n, st = G.stride[ax];
	if (st == 1) {
		// lines are contiguous: accumulate over j with the transposed matrix so the inner loop
		// runs over the n contiguous outputs of one line
		std::vector<double> MT((size_t) n * n);
		for (int i = 0; i < n; i++)
			for (int j = 0; j < n; j++) MT[(size_t) j * n + i] = M[(size_t) i * n + j];
		const int lines = G.nc / n;
		for (int r = 0; r < lines; r++) {
			double       *dst = out + (size_t) r * n;
			const double *src = in + (size_t) r * n;
			for (int i = 0; i < n; i++) dst[i] = 0;
			for (int j = 0; j < n; j++) {
				const double  w  = src[j];
				const double *mt = &MT[(size_t) j * n];
				for (int i = 0; i < n; i++) dst[i] += mt[i] * w;
			}
		}
		return;
	}
	const int outer = G.nc / (n * st); // blocks above the axis
	for (int o = 0; o < outer; o++) {
		for (int i = 0; i < n; i++) {
			double *dst = out + (size_t) o * n * st + (size_t) i * st;
			for (int x = 0; x < st; x++) dst[x] = 0;
			for (int j = 0; j < n; j++) {
				const double  w   = M[i * n + j];
				const double *src = in + (size_t) o * n * st + (size_t) j * st;
				for (int x = 0; x < st; x++) dst[x] += w * src[x];
			}
		}
	}
}
// ---- the same six transforms in O(n log n) (n a power of two): what the reference's DEFAULT patch solver does through FFTW's r2r
// plans (PatchSolvers/FftwPatchSolver.h:93-141: RODFT10/01, REDFT10/01, RODFT11, REDFT11; apps/3d/steady.cpp:126 --patch_solver fftw),
// where the dense products above are its DftPatchSolver (one dgemv per line). Own code (FFTW is not in the image): a radix-2 complex
// FFT over a batch of lines (every butterfly is a loop over contiguous lines: the compiler vectorises it) and the classic reductions
//   DCT-II : v_j = x_2j, v_{n-1-j} = x_{2j+1};  y_k = Re(e^{-i pi k / 2n} FFT_n(v)_k)                       (Makhoul 1980)
//   DCT-III: a_k = c_k x_k e^{+i pi k / 2n} (c_0 = 1/2); v = Re(conj-FFT_n(a)); y_2j = v_j, y_{2j+1} = v_{n-1-j}
//   DCT-IV : b_j = x_j e^{+i pi j / 2n} zero-padded to 2n; y_i = Re(e^{+i pi (2i+1) / 4n} conj-FFT_2n(b)_i)
//   DST-II(x)_i = DCT-II((-1)^j x_j)_{n-1-i};  DST-III(x)_i = (-1)^i DCT-III(reversed x)_i;  DST-IV(x)_i = (-1)^i DCT-IV(reversed x)_i
// with the matrices' own (unnormalised) scaling, so that either path can stand in transformAxis' place. Used by the TIMED baseline
// only (orc_set_fast_transforms, bench.py's cpu_baseline row "patch_solve_fft"); parity tests keep the dense products
// (tests/test_oracle_fast_transforms.py holds the two together to 1e-12).
int g_fast_transforms = 0;
struct FftTables {
	int                 n = 0;
	std::vector<int>    rev;
	std::vector<double> wr, wi; // e^{-2 pi i k / n}, k < n/2
};
const FftTables &fftTables(int n)
{
	static std::map<int, FftTables> cache;
	FftTables                      *t;
#pragma omp critical(orc_fft_tables)
	{
		auto it = cache.find(n);
		if (it == cache.end()) {
			FftTables T;
			T.n = n;
			T.rev.resize(n);
			int bits = 0;
			while ((1 << bits) < n) bits++;
			for (int i = 0; i < n; i++) {
				int r = 0;
				for (int b = 0; b < bits; b++)
					if (i & (1 << b)) r |= 1 << (bits - 1 - b);
				T.rev[i] = r;
			}
			T.wr.resize(n / 2 + 1), T.wi.resize(n / 2 + 1);
			for (int k = 0; k <= n / 2; k++) T.wr[k] = cos(2 * M_PI * k / n), T.wi[k] = -sin(2 * M_PI * k / n);
			it = cache.emplace(n, std::move(T)).first;
		}
		t = &it->second;
	}
	return *t;
}
// in-place DFT of length n over m lines: re/im[j * m + x]; sign -1: e^{-2 pi i jk/n}, +1: the conjugate kernel; unnormalised
void fftBatch(int n, int m, double *re, double *im, int sign)
{
	const FftTables &T = fftTables(n);
	for (int i = 0; i < n; i++) {
		const int r = T.rev[i];
		if (r > i)
			for (int x = 0; x < m; x++) std::swap(re[(size_t) i * m + x], re[(size_t) r * m + x]), std::swap(im[(size_t) i * m + x], im[(size_t) r * m + x]);
	}
	for (int len = 2; len <= n; len <<= 1) {
		const int half = len / 2, step = n / len;
		for (int base = 0; base < n; base += len)
			for (int k = 0; k < half; k++) {
				const double wr = T.wr[k * step], wi = sign < 0 ? T.wi[k * step] : -T.wi[k * step];
				double *ar = re + (size_t) (base + k) * m, *ai = im + (size_t) (base + k) * m;
				double *br = re + (size_t) (base + k + half) * m, *bi = im + (size_t) (base + k + half) * m;
				for (int x = 0; x < m; x++) {
					const double tr = br[x] * wr - bi[x] * wi, ti = br[x] * wi + bi[x] * wr;
					br[x] = ar[x] - tr, bi[x] = ai[x] - ti;
					ar[x] += tr, ai[x] += ti;
				}
			}
	}
}
// y[i * m + x] = sum_j M_t[i][j] x[j * m + x] for m lines at once (x, y: n x m, distinct); re, im: scratch of 2n x m each
void fastTransformBatch(DftType t, int n, int m, const double *x, double *y, double *re, double *im)
{
	const bool sine = (t == DST_II || t == DST_III || t == DST_IV);
	// the input a cosine transform of the same kind sees: sign-alternated (DST-II) or reversed (DST-III, DST-IV)
	auto in = [&](int j, int xx) -> double {
		if (t == DST_II) return (j & 1) ? -x[(size_t) j * m + xx] : x[(size_t) j * m + xx];
		if (sine) return x[(size_t) (n - 1 - j) * m + xx];
		return x[(size_t) j * m + xx];
	};
	if (t == DCT_II || t == DST_II) {
		for (int j = 0; j < n / 2; j++)
			for (int xx = 0; xx < m; xx++) {
				re[(size_t) j * m + xx]           = in(2 * j, xx);
				re[(size_t) (n - 1 - j) * m + xx] = in(2 * j + 1, xx);
			}
		std::fill(im, im + (size_t) n * m, 0.0);
		fftBatch(n, m, re, im, -1);
		for (int k = 0; k < n; k++) {
			const double c = cos(M_PI * k / (2.0 * n)), sn = sin(M_PI * k / (2.0 * n));
			double      *o = y + (size_t) (sine ? n - 1 - k : k) * m;
			for (int xx = 0; xx < m; xx++) o[xx] = re[(size_t) k * m + xx] * c + im[(size_t) k * m + xx] * sn;
		}
	} else if (t == DCT_III || t == DST_III) {
		for (int k = 0; k < n; k++) {
			const double c = (k == 0 ? 0.5 : 1.0) * cos(M_PI * k / (2.0 * n)), sn = (k == 0 ? 0.5 : 1.0) * sin(M_PI * k / (2.0 * n));
			for (int xx = 0; xx < m; xx++) {
				const double v            = in(k, xx);
				re[(size_t) k * m + xx] = v * c, im[(size_t) k * m + xx] = v * sn;
			}
		}
		fftBatch(n, m, re, im, +1);
		for (int j = 0; j < n / 2; j++)
			for (int xx = 0; xx < m; xx++) {
				const double a = re[(size_t) j * m + xx], b = re[(size_t) (n - 1 - j) * m + xx];
				y[(size_t) (2 * j) * m + xx]     = a;                                   // ((-1)^i = +1 on even rows)
				y[(size_t) (2 * j + 1) * m + xx] = sine ? -b : b;
			}
	} else { // DCT_IV, DST_IV
		for (int j = 0; j < n; j++) {
			const double c = cos(M_PI * j / (2.0 * n)), sn = sin(M_PI * j / (2.0 * n));
			for (int xx = 0; xx < m; xx++) {
				const double v            = in(j, xx);
				re[(size_t) j * m + xx] = v * c, im[(size_t) j * m + xx] = v * sn;
			}
		}
		std::fill(re + (size_t) n * m, re + (size_t) 2 * n * m, 0.0);
		std::fill(im + (size_t) n * m, im + (size_t) 2 * n * m, 0.0);
		fftBatch(2 * n, m, re, im, +1);
		for (int i = 0; i < n; i++) {
			const double ph = M_PI * (2 * i + 1) / (4.0 * n), c = cos(ph), sn = sin(ph), sg = (sine && (i & 1)) ? -1.0 : 1.0;
			for (int xx = 0; xx < m; xx++) y[(size_t) i * m + xx] = sg * (re[(size_t) i * m + xx] * c - im[(size_t) i * m + xx] * sn);
		}
	}
}
// transformAxis through the fast transforms: lines along an axis of stride > 1 are already a batch (the `st` contiguous cells);
// the x axis goes through a transpose of n x n tiles
void fastTransformAxis(const Geo &G, DftType t, int ax, const double *in, double *out)
{
	const int           n = G.n, st = G.stride[ax];
	constexpr int       MB = 64; // lines per batch: 2n x MB scratch stays in the first cache levels
	static thread_local std::vector<double> re, im, xin, yout; // (one set per OpenMP thread, grown once)
	if (re.size() < (size_t) 2 * n * MB) re.resize((size_t) 2 * n * MB), im.resize((size_t) 2 * n * MB), xin.resize((size_t) n * MB), yout.resize((size_t) n * MB);
	if (st == 1) {
		const int lines = G.nc / n;
		for (int l0 = 0; l0 < lines; l0 += MB) {
			const int m = std::min(MB, lines - l0);
			for (int x = 0; x < m; x++)
				for (int j = 0; j < n; j++) xin[(size_t) j * m + x] = in[(size_t) (l0 + x) * n + j];
			fastTransformBatch(t, n, m, xin.data(), yout.data(), re.data(), im.data());
			for (int x = 0; x < m; x++)
				for (int i = 0; i < n; i++) out[(size_t) (l0 + x) * n + i] = yout[(size_t) i * m + x];
		}
		return;
	}
	const int outer = G.nc / (n * st);
	for (int o = 0; o < outer; o++)
		for (int x0 = 0; x0 < st; x0 += MB) {
			const int     m   = std::min(MB, st - x0);
			const double *src = in + (size_t) o * n * st + x0;
			double       *dst = out + (size_t) o * n * st + x0;
			for (int j = 0; j < n; j++)
				for (int x = 0; x < m; x++) xin[(size_t) j * m + x] = src[(size_t) j * st + x];
			fastTransformBatch(t, n, m, xin.data(), yout.data(), re.data(), im.data());
			for (int i = 0; i < n; i++)
				for (int x = 0; x < m; x++) dst[(size_t) i * st + x] = yout[(size_t) i * m + x];
		}
}
struct SolvePlan {
	std::vector<double> fwd[3], inv[3], eig;
	DftType             ftype[3], itype[3];
};
SolvePlan makePlan(const orc_level *L, const Geo &G, int p)
{
	SolvePlan pl;
	const int n   = G.n;
	const int neu = L->neumann[p];
	// a side is Neumann for the solver only if it is a physical boundary with the bit set
	auto isNeu = [&](int s) { return ((neu >> s) & 1) && L->nbr_kind[(size_t) p * G.nsides + s] == 0; };
	pl.eig.assign(G.nc, 0.0);
	const double h = L->h[(size_t) p * G.dim]; // DftPatchSolver.h:148 uses spacings[0]
	for (int ax = 0; ax < G.dim; ax++) {
		bool lo = isNeu(2 * ax), hi = isNeu(2 * ax + 1);
		DftType f, i;
		if (lo && hi) {
			f = DCT_II;
			i = DCT_III;
		} else if (lo) {
			f = i = DCT_IV;
		} else if (hi) {
			f = i = DST_IV;
		} else {
			f = DST_II;
			i = DST_III;
		}
		pl.fwd[ax] = transformMatrix(f, n);
		pl.inv[ax] = transformMatrix(i, n);
		pl.ftype[ax] = f, pl.itype[ax] = i;
		// eigenvalues, FftwPatchSolver.h:143-168 == DftPatchSolver.h:150-166
		for (int c = 0; c < G.nc; c++) {
			int    xi = (c / G.stride[ax]) % n;
			double v;
			if (lo && hi)
				v = 4 / (h * h) * pow(sin(xi * M_PI / (2 * n)), 2);
			else if (lo || hi)
				v = 4 / (h * h) * pow(sin((xi + 0.5) * M_PI / (2 * n)), 2);
			else
				v = 4 / (h * h) * pow(sin((xi + 1) * M_PI / (2 * n)), 2);
			pl.eig[c] -= v;
		}
	}
	return pl;
}
// one plan per distinct (physical Neumann sides, spacing) — the reference's DomainK key
// (FftwPatchSolver.h:33-47) — built once per call, shared read-only by the patch loop
struct PlanCache {
	std::map<std::pair<int, double>, SolvePlan> plans;
	std::vector<const SolvePlan *>              of_patch;
	PlanCache(const orc_level *L, const Geo &G)
	{
		of_patch.resize(L->P);
		for (int p = 0; p < L->P; p++) {
			int key = 0;
			for (int s = 0; s < G.nsides; s++)
				if (((L->neumann[p] >> s) & 1) && L->nbr_kind[(size_t) p * G.nsides + s] == 0) key |= 1 << s;
			auto k  = std::make_pair(key, L->h[(size_t) p * G.dim]);
			auto it = plans.find(k);
			if (it == plans.end()) it = plans.emplace(k, makePlan(L, G, p)).first;
			of_patch[p] = &it->second;
		}
	}
};
void solvePatch(const orc_level *L, const Geo &G, const Ifaces &I, const PlanCache &PC, int p, const double *gamma,
                const double *f, double *u)
{
	const int           n = G.n;
	std::vector<double> a(f + (size_t) p * G.nc, f + (size_t) (p + 1) * G.nc), b(G.nc);
	// f_copy -= 2/h^2 * gamma on every face that has a neighbour (DftPatchSolver.h:190-202)
	for (int s = 0; s < G.nsides; s++) {
		size_t fidx = (size_t) p * G.nsides + s;
		if (L->nbr_kind[fidx] == 0) continue;
		double h2 = pow(L->h[(size_t) p * G.dim + s / 2], 2);
		for (int bb = 0; bb < (G.dim == 3 ? n : 1); bb++)
			for (int aa = 0; aa < n; aa++)
				a[G.faceCell(s, aa, bb, 0)] -= 2.0 / h2 * gamma[(size_t) I.own[fidx] * G.nf + aa + n * bb];
	}
	const SolvePlan &pl = *PC.of_patch[p];
	double   *src = a.data(), *dst = b.data();
	const bool fast = g_fast_transforms && (n & (n - 1)) == 0 && n >= 4;
	for (int ax = 0; ax < G.dim; ax++) {
		if (fast)
			fastTransformAxis(G, pl.ftype[ax], ax, src, dst);
		else
			transformAxis(G, pl.fwd[ax], ax, src, dst);
		std::swap(src, dst);
	}
	for (int c = 0; c < G.nc; c++) src[c] /= pl.eig[c];
	bool all_neu = true;
	for (int s = 0; s < G.nsides; s++)
		if (!(((L->neumann[p] >> s) & 1) && L->nbr_kind[(size_t) p * G.nsides + s] == 0)) all_neu = false;
	// reference tests neumann.all() (DftPatchSolver.h:208); identical for a 1-patch domain
	if (all_neu) src[0] = 0;
	for (int ax = 0; ax < G.dim; ax++) {
		if (fast)
			fastTransformAxis(G, pl.itype[ax], ax, src, dst);
		else
			transformAxis(G, pl.inv[ax], ax, src, dst);
		std::swap(src, dst);
	}
	double scale = pow(2.0 / n, G.dim);
	double *up   = u + (size_t) p * G.nc;
	for (int c = 0; c < G.nc; c++) up[c] = src[c] * scale;
}

// diagonal of the assembled operator seen from cell c of patch p (used by the product's
// pointwise smoothers only)
inline double faceDiagCoef(int kind, bool neu)
{
	switch (kind) {
		case 1: return 2.0;
		case 2: return 0; // filled by caller (dim dependent)
		case 3: return 0;
		default: return neu ? 1.0 : 3.0;
	}
}
} // namespace

extern "C" {
void orc_set_threads(int nthreads) { g_threads = nthreads < 1 ? 1 : nthreads; }
void orc_set_fast_transforms(int on) { g_fast_transforms = on != 0; }

int orc_num_ifaces(const orc_level *L) { return buildIfaces(L).count; }

void orc_iface_index(const orc_level *L, int32_t *iface_index)
{
	Ifaces I = buildIfaces(L);
	for (size_t i = 0; i < I.own.size(); i++) iface_index[i] = I.own[i];
}

void orc_interp(const orc_level *L, const double *u, double *gamma)
{
	Geo    G(L);
	Ifaces I = buildIfaces(L);
	memset(gamma, 0, sizeof(double) * (size_t) I.count * G.nf);
	for (int p = 0; p < L->P; p++) interpPatch(L, G, I, p, u, gamma);
}

void orc_apply_with_gamma(const orc_level *L, const double *u, const double *gamma, double *f)
{
	Geo    G(L);
	Ifaces I = buildIfaces(L);
#pragma omp parallel for num_threads(g_threads) schedule(static)
	for (int p = 0; p < L->P; p++) applyPatch(L, G, &I, p, u, gamma, f, true);
}

void orc_apply(const orc_level *L, const double *u, double *f)
{
	Geo                 G(L);
	Ifaces              I = buildIfaces(L);
	std::vector<double> gamma((size_t) I.count * G.nf, 0.0);
	for (int p = 0; p < L->P; p++) interpPatch(L, G, I, p, u, gamma.data());
#pragma omp parallel for num_threads(g_threads) schedule(static)
	for (int p = 0; p < L->P; p++) applyPatch(L, G, &I, p, u, gamma.data(), f, true);
}

void orc_patch_apply(const orc_level *L, const double *u, double *f)
{
	Geo G(L);
	for (int p = 0; p < L->P; p++) applyPatch(L, G, nullptr, p, u, nullptr, f, false);
}

void orc_add_iface_rhs(const orc_level *L, const double *gamma, double *f)
{
	Geo    G(L);
	Ifaces I = buildIfaces(L);
	for (int p = 0; p < L->P; p++)
		for (int s = 0; s < G.nsides; s++) {
			size_t fidx = (size_t) p * G.nsides + s;
			if (L->nbr_kind[fidx] == 0) continue;
			// StarPatchOp.h:195 takes spacings[s.axis()] where axis() returns bool (Side.h:105);
			// identical for the cubic cells every config uses. We use the true axis.
			double h2 = pow(L->h[(size_t) p * G.dim + s / 2], 2);
			for (int b = 0; b < (G.dim == 3 ? G.n : 1); b++)
				for (int a = 0; a < G.n; a++)
					f[(size_t) p * G.nc + G.faceCell(s, a, b, 0)]
					-= 2.0 / h2 * gamma[(size_t) I.own[fidx] * G.nf + a + G.n * b];
		}
}

void orc_patch_solve(const orc_level *L, const double *gamma, const double *f, double *u)
{
	Geo       G(L);
	Ifaces    I = buildIfaces(L);
	PlanCache PC(L, G);
#pragma omp parallel for num_threads(g_threads) schedule(dynamic)
	for (int p = 0; p < L->P; p++) solvePatch(L, G, I, PC, p, gamma, f, u);
}

void orc_smooth(const orc_level *L, const double *f, double *u)
{
	Geo                 G(L);
	Ifaces              I = buildIfaces(L);
	std::vector<double> gamma((size_t) I.count * G.nf, 0.0);
	for (int p = 0; p < L->P; p++) interpPatch(L, G, I, p, u, gamma.data());
	PlanCache PC(L, G);
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 1)
	for (int p = 0; p < L->P; p++) solvePatch(L, G, I, PC, p, gamma.data(), f, u);
}

// PatchSolvers/BiCGStabSolver.h:114-132 under SchurHelper::solveWithInterface's loop (SchurHelper.h:318-331), the smoother the
// 2D driver builds with --patch_solver bcgs (apps/2d/steady.cpp:326-327): gamma from the old iterate, then per patch
// f_copy = f - interface terms (addInterfaceToRHS) and BiCGStab<D>::solve (BiCGStab.h:45-106, no preconditioner) with
// StarPatchOp::apply as the operator, the patch's current values as the initial guess. its[p] = iterations of patch p (may be null).
static double g_bcgs_tol    = 1e-12; // BiCGStabSolver(op, tol = 1e-12, max_it = 1000), BiCGStabSolver.h:103-108: what a cycle
static int    g_bcgs_max_it = 1000;  // with smoother == 3 uses
void orc_set_patch_bcgs(double tol, int max_it)
{
	g_bcgs_tol    = tol;
	g_bcgs_max_it = max_it;
}
void orc_smooth_bcgs(const orc_level *L, const double *f, double *u, double tol, int max_it, int *its)
{
	Geo                 G(L);
	Ifaces              I = buildIfaces(L);
	const size_t        N = (size_t) L->P * G.nc;
	std::vector<double> gamma((size_t) I.count * G.nf, 0.0);
	for (int p = 0; p < L->P; p++) interpPatch(L, G, I, p, u, gamma.data());
	std::vector<double> b(f, f + N);
	orc_add_iface_rhs(L, gamma.data(), b.data());
	// level-sized scratch: patch p only ever touches its own n^D slot of each
	std::vector<double> resid(N), rhat(N), pv(N), ap(N), s(N), as(N);
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 1)
	for (int p = 0; p < L->P; p++) {
		const size_t o0 = (size_t) p * G.nc, o1 = o0 + G.nc;
		auto dot = [&](const std::vector<double> &a, const std::vector<double> &c) {
			double t = 0;
			for (size_t i = o0; i < o1; i++) t += a[i] * c[i];
			return t;
		};
		auto A = [&](const double *x, double *y) { applyPatch(L, G, nullptr, p, x, nullptr, y, false); };
		A(u, resid.data());
		for (size_t i = o0; i < o1; i++) resid[i] = -1 * resid[i] + b[i];
		const double r0_norm = sqrt(dot(resid, resid));
		for (size_t i = o0; i < o1; i++) rhat[i] = pv[i] = resid[i];
		double rho     = dot(rhat, resid);
		int    num_its = 0;
		while (sqrt(dot(resid, resid)) / r0_norm > tol && num_its < max_it) {
			A(pv.data(), ap.data());
			const double alpha = rho / dot(rhat, ap);
			for (size_t i = o0; i < o1; i++) s[i] = resid[i] + ap[i] * -alpha;
			A(s.data(), as.data());
			const double omega = dot(as, s) / dot(as, as);
			for (size_t i = o0; i < o1; i++) u[i] += pv[i] * alpha + s[i] * omega;
			for (size_t i = o0; i < o1; i++) resid[i] += ap[i] * -alpha + as[i] * -omega;
			const double rho_new = dot(resid, rhat);
			const double beta    = rho_new * alpha / (rho * omega);
			for (size_t i = o0; i < o1; i++) pv[i] += ap[i] * -omega;
			for (size_t i = o0; i < o1; i++) pv[i] = beta * pv[i] + resid[i];
			num_its++;
			rho = rho_new;
		}
		if (its) its[p] = num_its;
	}
}

void orc_restrict(const orc_level *fine, const orc_level *coarse, const double *fv, double *cv)
{
	Geo       G(fine);
	const int n = G.n, nz = (G.dim == 3) ? n : 1;
	memset(cv, 0, sizeof(double) * (size_t) coarse->P * G.nc);
	// fine patches in ascending id order (std::set<ILCFineToCoarseMetadata>, InterLevelComm.h:47-50)
	std::vector<std::pair<int, int>> order;
	for (int p = 0; p < fine->P; p++) order.emplace_back(fine->id[p], p);
	std::sort(order.begin(), order.end());
	for (auto &op : order) {
		int           p  = op.second;
		const double *fp = fv + (size_t) p * G.nc;
		double       *cp = cv + (size_t) fine->parent[p] * G.nc;
		int           o  = fine->orth_on_parent[p];
		if (o >= 0) {
			int st[3] = {(o & 1) ? n : 0, (o & 2) ? n : 0, (o & 4) ? n : 0};
			for (int z = 0; z < nz; z++)
				for (int y = 0; y < n; y++)
					for (int x = 0; x < n; x++) {
						int ci = (x + st[0]) / 2 + n * ((y + st[1]) / 2);
						if (G.dim == 3) ci += n * n * ((z + st[2]) / 2);
						cp[ci] += fp[x + n * y + n * n * z] / (1 << G.dim);
					}
		} else {
			for (int c = 0; c < G.nc; c++) cp[c] += fp[c];
		}
	}
}

void orc_prolong_add(const orc_level *fine, const orc_level *coarse, const double *cv, double *fv)
{
	(void) coarse;
	Geo       G(fine);
	const int n = G.n, nz = (G.dim == 3) ? n : 1;
	for (int p = 0; p < fine->P; p++) {
		double       *fp = fv + (size_t) p * G.nc;
		const double *cp = cv + (size_t) fine->parent[p] * G.nc;
		int           o  = fine->orth_on_parent[p];
		if (o >= 0) {
			int st[3] = {(o & 1) ? n : 0, (o & 2) ? n : 0, (o & 4) ? n : 0};
			for (int z = 0; z < nz; z++)
				for (int y = 0; y < n; y++)
					for (int x = 0; x < n; x++) {
						int ci = (x + st[0]) / 2 + n * ((y + st[1]) / 2);
						if (G.dim == 3) ci += n * n * ((z + st[2]) / 2);
						fp[x + n * y + n * n * z] += cp[ci];
					}
		} else {
			for (int c = 0; c < G.nc; c++) fp[c] += cp[c];
		}
	}
}

// ------------------------------------------------------------------------------------------
// Product smoothers restated (not reference functions).
// Ghost value on a neighbour face = 2*gamma - own face cell.
// ------------------------------------------------------------------------------------------
static void cellCoefs(const orc_level *L, const Geo &G, int p, int x, int y, int z, double *diag,
                      const double *up, const double *ghost /*[nsides][nf]*/, double *offsum)
{
	// returns diag = sum_a k_a / h_a^2 (positive), offsum = sum_a (lo + hi)/h_a^2 excluding
	// the cell itself and physical-boundary ghosts
	const int n  = G.n;
	int       c[3] = {x, y, z};
	double    d = 0, o = 0;
	for (int ax = 0; ax < G.dim; ax++) {
		double h2 = L->h[(size_t) p * G.dim + ax];
		h2 *= h2;
		double k   = 2.0;
		int    idx = x + n * y + n * n * z;
		int    fa[2] = {0, 0}, kk = 0;
		for (int i = 0; i < G.dim; i++)
			if (i != ax) fa[kk++] = c[i];
		for (int side = 0; side < 2; side++) {
			int  s      = 2 * ax + side;
			bool atface = side ? (c[ax] == n - 1) : (c[ax] == 0);
			if (!atface) {
				o += up[idx + (side ? G.stride[ax] : -G.stride[ax])] / h2;
			} else {
				int kind = L->nbr_kind[(size_t) p * G.nsides + s];
				if (kind == 0) {
					k += ((L->neumann[p] >> s) & 1) ? -1.0 : 1.0;
				} else {
					o += ghost[(size_t) s * G.nf + fa[0] + n * fa[1]] / h2;
				}
			}
		}
		d += k / h2;
	}
	*diag   = d;
	*offsum = o;
}

static void ghostsFromGamma(const orc_level *L, const Geo &G, const Ifaces &I, int p,
                            const double *u, const double *gamma, double *ghost)
{
	const int n = G.n;
	for (int s = 0; s < G.nsides; s++) {
		size_t f = (size_t) p * G.nsides + s;
		if (L->nbr_kind[f] == 0) continue;
		for (int b = 0; b < (G.dim == 3 ? n : 1); b++)
			for (int a = 0; a < n; a++)
				ghost[(size_t) s * G.nf + a + n * b]
				= 2 * gamma[(size_t) I.own[f] * G.nf + a + n * b] - u[(size_t) p * G.nc + G.faceCell(s, a, b, 0)];
	}
}

void orc_jacobi(const orc_level *L, const double *f, double *u, double omega)
{
	// u <- u + omega * D^-1 (f - A u), D = diagonal of the assembled operator. On coarse/fine
	// faces the ghost depends on the cell itself: fine side d(ghost)/d(cell) = 5/6 (3D) or 5/6
	// (2D: 2*5/6-1 = 2/3), coarse side -1/3; folded into the diagonal below.
	Geo                 G(L);
	Ifaces              I = buildIfaces(L);
	std::vector<double> au((size_t) L->P * G.nc);
	orc_apply(L, u, au.data());
	const int n = G.n, nz = (G.dim == 3) ? n : 1;
	for (int p = 0; p < L->P; p++) {
		for (int z = 0; z < nz; z++)
			for (int y = 0; y < n; y++)
				for (int x = 0; x < n; x++) {
					int    c[3] = {x, y, z};
					double d    = 0;
					for (int ax = 0; ax < G.dim; ax++) {
						double h2 = L->h[(size_t) p * G.dim + ax];
						h2 *= h2;
						double k = 2.0;
						for (int side = 0; side < 2; side++) {
							int  s      = 2 * ax + side;
							bool atface = side ? (c[ax] == n - 1) : (c[ax] == 0);
							if (!atface) continue;
							int kind = L->nbr_kind[(size_t) p * G.nsides + s];
							if (kind == 0)
								k += ((L->neumann[p] >> s) & 1) ? -1.0 : 1.0;
							else if (kind == 2)
								k -= (G.dim == 3) ? 5.0 / 6.0 : 2.0 / 3.0;
							else if (kind == 3)
								k += 1.0 / 3.0;
						}
						d += k / h2;
					}
					size_t i = (size_t) p * G.nc + x + n * y + n * n * z;
					// A's diagonal is -d
					u[i] += omega * (f[i] - au[i]) / (-d);
				}
	}
}

void orc_patch_rbgs(const orc_level *L, const double *f, double *u)
{
	Geo                 G(L);
	Ifaces              I = buildIfaces(L);
	std::vector<double> gamma((size_t) I.count * G.nf, 0.0);
	for (int p = 0; p < L->P; p++) interpPatch(L, G, I, p, u, gamma.data());
	const int n = G.n, nz = (G.dim == 3) ? n : 1;
#pragma omp parallel for num_threads(g_threads) schedule(static)
	for (int p = 0; p < L->P; p++) {
		std::vector<double> ghost((size_t) G.nsides * G.nf, 0.0);
		ghostsFromGamma(L, G, I, p, u, gamma.data(), ghost.data());
		double *up = u + (size_t) p * G.nc;
		for (int colour = 0; colour < 2; colour++)
			for (int z = 0; z < nz; z++)
				for (int y = 0; y < n; y++)
					for (int x = 0; x < n; x++) {
						if (((x + y + z) & 1) != colour) continue;
						double d, o;
						cellCoefs(L, G, p, x, y, z, &d, up, ghost.data(), &o);
						int i = x + n * y + n * n * z;
						up[i] = (o - f[(size_t) p * G.nc + i]) / d;
					}
	}
}

// ---- a1 ------------------------------------------------------------------------------------
static void smoothLevel(const orc_level *L, const orc_cycle_opts *o, bool coarsest,
                        const double *f, double *u)
{
	if (o->smoother == 0 || (coarsest && o->exact_coarse && L->P == 1))
		orc_smooth(L, f, u);
	else if (o->smoother == 1)
		orc_jacobi(L, f, u, o->omega);
	else if (o->smoother == 3)
		orc_smooth_bcgs(L, f, u, g_bcgs_tol, g_bcgs_max_it, nullptr);
	else
		orc_patch_rbgs(L, f, u);
}
static void visit(const orc_level *levels, int nlevels, int l, const orc_cycle_opts *o,
                  const double *f, double *u)
{
	const orc_level *L = &levels[l];
	Geo              G(L);
	size_t           N        = (size_t) L->P * G.nc;
	bool             coarsest = (l == nlevels - 1);
	if (coarsest) {
		for (int i = 0; i < o->coarse_sweeps; i++) smoothLevel(L, o, true, f, u);
		return;
	}
	auto descend = [&]() {
		// prepCoarser, Cycle.h:56-68
		std::vector<double> r(N);
		orc_apply(L, u, r.data());
		for (size_t i = 0; i < N; i++) r[i] = -1 * r[i] + f[i];
		size_t              Nc = (size_t) levels[l + 1].P * G.nc;
		std::vector<double> cu(Nc, 0.0), cf(Nc, 0.0);
		orc_restrict(L, &levels[l + 1], r.data(), cf.data());
		visit(levels, nlevels, l + 1, o, cf.data(), cu.data());
		// prepFiner, Cycle.h:74-80
		orc_prolong_add(L, &levels[l + 1], cu.data(), u);
	};
	for (int i = 0; i < o->pre_sweeps; i++) smoothLevel(L, o, false, f, u);
	descend();
	if (o->cycle_type == 1) {
		for (int i = 0; i < o->mid_sweeps; i++) smoothLevel(L, o, false, f, u);
		descend();
	}
	for (int i = 0; i < o->post_sweeps; i++) smoothLevel(L, o, false, f, u);
}
void orc_cycle(const orc_level *levels, int nlevels, const orc_cycle_opts *o, const double *f,
               double *u)
{
	Geo    G(&levels[0]);
	size_t N = (size_t) levels[0].P * G.nc;
	memset(u, 0, sizeof(double) * N); // Cycle.h:118
	visit(levels, nlevels, 0, o, f, u);
}

int orc_bicgstab(const orc_level *levels, int nlevels, const orc_cycle_opts *o, int use_prec,
                 const double *b, double *x, int max_it, double tol, double *final_rel_resid)
{
	const orc_level *L = &levels[0];
	Geo              G(L);
	const size_t     N = (size_t) L->P * G.nc;
	auto dot  = [&](const std::vector<double> &a, const std::vector<double> &c) {
        double s = 0;
        for (size_t i = 0; i < N; i++) s += a[i] * c[i];
        return s;
	};
	auto norm = [&](const std::vector<double> &a) { return sqrt(dot(a, a)); };
	std::vector<double> resid(N), ms(N), mp(N), rhat, p, ap(N), as(N), s(N);
	// BiCGStab.h:57-69
	orc_apply(L, x, resid.data());
	for (size_t i = 0; i < N; i++) resid[i] = -1 * resid[i] + b[i];
	double r0_norm = norm(resid);
	rhat           = resid;
	p              = resid;
	double rho     = dot(rhat, resid);
	int    num_its = 0;
	while (norm(resid) / r0_norm > tol && num_its < max_it) {
		if (use_prec) {
			orc_cycle(levels, nlevels, o, p.data(), mp.data());
			orc_apply(L, mp.data(), ap.data());
		} else {
			orc_apply(L, p.data(), ap.data());
		}
		double alpha = rho / dot(rhat, ap);
		for (size_t i = 0; i < N; i++) s[i] = resid[i] + ap[i] * -alpha;
		if (use_prec) {
			orc_cycle(levels, nlevels, o, s.data(), ms.data());
			orc_apply(L, ms.data(), as.data());
		} else {
			orc_apply(L, s.data(), as.data());
		}
		double omega = dot(as, s) / dot(as, as);
		const std::vector<double> &dp = use_prec ? mp : p, &ds = use_prec ? ms : s;
		for (size_t i = 0; i < N; i++) x[i] += dp[i] * alpha + ds[i] * omega;
		for (size_t i = 0; i < N; i++) resid[i] += ap[i] * -alpha + as[i] * -omega;
		double rho_new = dot(resid, rhat);
		double beta    = rho_new * alpha / (rho * omega);
		for (size_t i = 0; i < N; i++) p[i] += ap[i] * -omega;
		for (size_t i = 0; i < N; i++) p[i] = beta * p[i] + resid[i];
		num_its++;
		rho = rho_new;
	}
	if (final_rel_resid) *final_rel_resid = norm(resid) / r0_norm;
	return num_its;
}
} // extern "C"
