// HO kernel, second generation: element batching.
//
// du = M^-1 (K_vol + K_face) u, lumped mass,
// element extrema -- but organised for wave64 occupancy of the "column" phases:
//
//   * a workgroup of 256 threads (4 wavefronts) works on NB = floor(256 / Q^2) elements at once
//     (p = 3: Q = 6, 36 quadrature columns per element, NB = 7 -> 252 of 256 lanes busy; the
//     one-element-per-wavefront kernel keeps 36 of 64 busy);
//   * thread roles per phase:  column (eb, qx, qy)  |  face row (eb, face, q1)  |  dof (eb, i)
//     | flat task loops over small intermediates;
//   * the 1-D basis tables are read from global memory with compile-time indices, i.e. through
//     the scalar cache into SGPRs and used as scalar FMA operands; only lane-dependent rows sit
//     in VGPRs.  LDS holds element data only;
//   * faces are sum-factorised (one thread per face row of quadrature points, geometry from the
//     9 face nodes) instead of evaluated point by point;
//   * the right-hand side is tested directly with the Gauss-Legendre nodal basis (the basis of
//     the local mass solve), so no forward change of basis is needed; PCG vectors live in
//     registers of the dof threads, only the search direction goes through LDS.
//
// Reference semantics: LocalInverseHOSolver::CalcHOSolution PA branch (remhos_ho.cpp:119-128),
// K_HO = ConvectionIntegrator + transposed DGTraceIntegrator (remhos.cpp:646-678), DGMassInverse
// (remhos_ho.cpp:79-80), lumped mass M_HO*1 (remhos.cpp:1632).
#pragma once
#include "rmh_kernels.hpp"
#include <type_traits>

namespace rmh
{

// The compile-time switches that remain are the ones whose value differs per order or per kernel, each with its measured
// effect; variants that were built, measured and rejected (persistent workgroups with register prefetch, wave priorities,
// traces stored as values, per-dof back-transform and x-leg, whole second wavefront for the 17 extra columns of p = 6, ...)
// are kept as patches with their numbers under tools/experiments/, not as dead branches of the product kernel.
//
// adj(J) v and det J through two cross products in the column phase (24 instead of 30 operations per quadrature point).  Round 3:
// a gain at p = 3 only (at p = 6 the different register lifetimes cost 8 %); re-measured with the hierarchical nodes of round 4:
// p = 4, 5 +1.0 %, p = 6 +0.2 %, p = 2 -0.5 %
#ifndef RMH_ADJ_CROSS
#define RMH_ADJ_CROSS (P >= 3)
#endif
// fresh views of the constant table per unrolled quadrature plane from this order on (see tab_view)
#ifndef RMH_VIEW_MINP
#define RMH_VIEW_MINP 4
#endif
// lo 4 stage: the RD solver's z = K_vol u needs the Bernstein test basis; its x-direction comes out of the x-leg of phase G
// as a second accumulator (one conversion leg less) -- not at p = 3, where that accumulator is the register that spills
#ifndef RMH_RD_XLEG
#define RMH_RD_XLEG (P >= 4)
#endif
// y-leg of the test contractions in place: less LDS per workgroup, more workgroups per CU (p = 6: 30 -> 21 KB, the sixth
// workgroup; p = 5: 11.5 k -> 12.7 k MDOFs*stage/s, p = 4: 13.4 k -> 14.1 k; p = 3: -1.3 %, the registers limit it anyway).
// The lo 4 kernels use it at every order: it is what lets 7 elements share a workgroup at p = 3 (K2Cfg).
#ifndef RMH_INPLACE_Y
#define RMH_INPLACE_Y (P >= 4)
#endif

// the lo 5 stage kernel forms the x-contraction of u and the face jumps in its load phase (ho_kernel2, ULN): u itself is not in LDS
#ifndef RMH_ULN
#define RMH_ULN (P >= 3)
#endif
// hierarchical directions of the Q2 mesh nodes (bit mask; explained where the kernel uses it, below)
#ifndef RMH_HIER
#define RMH_HIER 5
#endif

// (NOU: no u slot in the work region -- the whole-stage kernel of lo 5 where RMH_ULN holds, see K2For)
template <int P, bool LO4 = false, bool BOTH = false, bool NOU = false>
struct K2Cfg : TabLayout<P>
{
   using T = TabLayout<P>;
   static constexpr int D = T::D, Q = T::Q;
   static constexpr int D2 = D * D, D3 = D * D * D, Q2 = Q * Q;
   // threads per workgroup: 256 (4 wavefronts, one per SIMD) at p <= 3, where several elements fill the lanes of the
   // column phases.  p = 4, 5: ONE wavefront and one element per workgroup (Q^2 = 49 and 64 columns on 64 lanes): no
   // barrier synchronises more than a wavefront, the element sums are a thread-local sum + one DPP wave sum
   // (p = 4, -rs 4: 12.4 k -> 13.4 k MDOFs*stage/s; p = 5: 9.6 k -> 11.6 k).  p = 6 (81 columns): two wavefronts, one element
   // (more elements per workgroup at p >= 4 was measured in every shape the LDS admits: -7 ... -53 %).
   static constexpr int NT = (P == 6) ? 128 : ((P == 5 || P == 4) ? 64 : 256);
   // elements per workgroup at p <= 3: as many as fill the 256 lanes in the column phases -- p = 3: 7 (252 lanes; on 128- or
   // 192-thread workgroups with 3 / 5 elements -10 % / -4 %) -- but at p = 2 9, not the 10 whose columns fit: their 243 dofs
   // take ONE round of the dof role where 270 take two (-rs 5: 9.3 k -> 11.2 k MDOFs*stage/s, bit-identical)
   static constexpr int NB = (P >= 4) ? 1 : ((P == 2) ? (NT / Q2 < NT / D3 ? NT / Q2 : NT / D3) : NT / Q2);
   // (HO + RD in one kernel used to carry one element less -- 6 at p = 3 -- for the LDS of the RD extras; with the layout below
   // all 7 fit at three workgroups per CU: lo 4 at p = 3 12.4 k -> 13.1 k MDOFs*stage/s)
   static constexpr int DR = (NB * D3 + NT - 1) / NT; // dof rounds per thread
   // per-element LDS block (doubles): a work region W whose contents change with the phase, and
   // the face buffer.
   //   phases A-C : [X(t),V nodes 162 (or their x-contractions, XPK) | u D3 | neighbour traces 6 D2 | U1 2 Q S2]
   //   phases C-G : [R3 3 Q2 D | R2 3 Q D2]
   //   PCG, J     : [sA D3 | M1 / R2' Q S2 | R3' Q2 D | sB D3]
   static constexpr int cmax(int a, int b) { return a > b ? a : b; }
   // Strides and the LDS banks (round 4, tools/pmc_variants.sh with builds that end at a phase mark: SQ_LDS_BANK_CONFLICT of
   // the p = 3 stage by phase -- face rows 30 %, y-leg 21 %, back-transform 18 %, R3 stores 13 %, column pass 12 %).  S2: the
   // odd row stride of U1 / M1 keeps the y-back stores of the PCG conflict-free (an even stride lets the column pass read with
   // ds_read_b128: -6 % LDS cycles, +16 % conflicts elsewhere, +-0 in time).  Padding the six trace blocks (D^2 + 2) and other
   // element strides (== 8 or 24 mod 32) took 13-25 % of the conflicts away at -0.5 ... +0.2 % in time; the face-major order of
   // the face rows (ho_kernel2) takes 23 % at +0.25 % and needs the element stride == 2 (mod 32).
   // (round 5: "odd" must hold for the stride itself -- D^2 + 1 is EVEN at odd D (p = 2, 4, 6), and then the column pass's
   // ds_read2_b64 of a U1 / M1 row (banks mod 32 dwords, 16-lane groups) sends lanes qx = 0 and qx = 8 to the same banks: p = 6
   // SQ_LDS_BANK_CONFLICT 1.10e8 -> 4.4e7 per launch (-60 %), LDS-array cycles -11 %, 21.7 k -> 22.1 k MDOFs*stage/s (+2.0 %),
   // lo 4 at p = 6 +1.2 %; p = 2, 4 +-0; D^2 + 4 the same as D^2 + 2.  Bit-identical.)
#ifndef RMH_S2PAD
#define RMH_S2PAD ((D2 & 1) ? 2 : 1)
#endif
   static constexpr int S2 = D2 + (RMH_S2PAD); // padded row stride of U1 / M1: odd
   // X-CONTRACTED MESH NODES (round 6).  The Q columns of a plane qx all contract the 27 nodes of X(t) and V along x with the same
   // three 1-D basis values: 162 of the 270 FP64 operations a column spends on the nodes, done Q times over.  With XPK != 0 the
   // thread that LOADS a node line (comp, ay, az) -- three nodes of x0 and of v -- contracts it for all Q planes on its way to
   // LDS (phase A: 6 Q FMAs per line, in the shadow of the memory round trip), and the column threads read the values of their qx:
   //   XPK = 2: [xl | vl][qx][line] + the raw hierarchical nodes x1, x2 of every line (the x-derivative xd = dL1 x1 + dL2 x2 stays
   //            with the column: 2 Q 27 + 54 doubles instead of 162 -- what the p = 3 work region has room for in phases A-C);
   //   XPK = 3: [xl | vl | xd][qx][line], no raw nodes (3 Q 27 doubles: p = 6, where the registers, not the LDS, cap the occupancy).
   //   (p = 3, xd of the components 0 and 1 as well -- 18 of the 27 lines fit the stage kernel's freed u slot without a byte more LDS, 36 more
   //   FP64 operations off a column: the compiler then spills in the column pass, 168 VGPRs + 28 B/lane of scratch, 22.2 -> 21.7 k; with
   //   scheduling fences 144 B/lane.  Not kept.)
   // Same operations in the same order as in the column: bit-identical.  Needs the hierarchical x form (RMH_HIER & 1).
   // Measured (tools/kbench.py, one box): p = 3 21 768 -> 22 335 / 22 445 (+2.6 ... 3.1 %), p = 6 25 440 -> 25 867 (+1.7 %); static FP64
   // instructions of the column phase 796 -> 688 (p = 3), 2006 -> 1682 (p = 6, both column forms), its LDS reads 108 -> 80 / 357 -> 273.
   // p = 4: XPK = 2 (HO kernel) / 3 (stage kernel, whose u slot is free: K2For) -- 12.6 KB per one-wavefront workgroup, still twelve per CU
   // (the LDS is allocated in finer granules than K2Cfg::LDS_BYTES assumes): cube01_hex -rs 5 23 500 -> 23 900 (XPK = 2) -> 24 090 (+2.5 %).
   // Not elsewhere: the lo 4 kernels of p = 6 +-0 (their LDS grows by two granules), those of p = 3 have no room; p = 5 with only vl
   // precomputed (all its LDS admits): -5 % (the tenth workgroup per CU; p = 4 with vl alone: +0.2 %).
#ifndef RMH_XPK
#ifndef RMH_XPK4
#define RMH_XPK4 (NOU ? 3 : 2)
#endif
#define RMH_XPK ((RMH_HIER & 1) ? ((P == 3 && !LO4) ? 2 : ((P == 6 && !LO4) ? 3 : ((P == 4 && !LO4) ? RMH_XPK4 : 0))) : 0)
#endif
   static constexpr int XPK = RMH_XPK;
   static constexpr int XVN = XPK == 0 ? 162 : ((XPK == 2 ? 2 * 27 * Q + 54 : 3 * 27 * Q) + 1) / 2 * 2; // (even: u stays 16-byte aligned)
   static constexpr int oXV = 0, oXR = 2 * 27 * Q, oU = XVN, oNb = oU + (NOU ? 0 : D3), oU1 = oNb + 6 * D2, PA = oU1 + 2 * Q * S2;
   // test tensors: r = 0 rhs (GL basis; Bernstein in the RD-only kernel), 1 lumped mass, 2 Jacobi diagonal.
   // When HO and RD run in the same kernel, z = K_vol u in the Bernstein basis is obtained from the GL-tested
   // volume rhs by the 1-D change of test basis Cf along y and z (phi^B_i = sum_k C[k][i] l_k; the x-leg of phase G
   // tests with both bases), not by a fourth tensor through phases C-G.
   static constexpr int NR = 3;
   // INPLACE_Y: the y-leg writes its D outputs over the first D of the Q inputs of its own line (R2 inside R3)
   static constexpr bool INPLACE_Y = RMH_INPLACE_Y || LO4;
   // R3 = [r][qy][qx][iz]: QYS = stride of qy, R3S = tensor stride.  In place (R2 inside R3) the dof threads of the x-leg read
   // R2[(jx, iy, iz)] with their lanes iy QYS doubles apart: Q D = 24 doubles = 48 dwords at p = 3 puts iy = 0, 2 and iy = 1, 3
   // into the same banks -- 2-way conflicts on all 18 reads of a dof (tools/pmc_variants.sh, round 6: 6.4e7 of the lo 4 stage's
   // 2.15e8 SQ_LDS_BANK_CONFLICT per launch sat in that leg).  Multi-element workgroups pad the line by RMH_RYPAD doubles; the
   // work region has the room (PA > PF).
   // (where Q D is a multiple of 8 doubles -- p = 3, p = 1; p = 2 has 15: its three iy already fall into different banks)
#ifndef RMH_RYPAD
#define RMH_RYPAD 1
#endif
   static constexpr int RYPAD = (INPLACE_Y && NB > 1 && (Q * D) % 8 == 0) ? RMH_RYPAD : 0;
   static constexpr int QYS = Q * D + RYPAD, R3S = Q * QYS;
   // R2 = [r][jx][iy + D iz] (not in place): R2S = stride of jx.  The y-leg's tasks (jx, iz) store their D outputs 8 iz + 2 R2S jx
   // dwords apart: with R2S = D^2 = 16 doubles (p = 3) the four jx of a 16-lane group fall into the same banks -- 4-way conflicts
   // on every store of the leg (round 4's attribution: the y-leg held a fifth of the p = 3 stage's bank conflicts).  One double
   // of padding per jx spreads them over all banks; the x-leg reads jx R2S + i2 with consecutive i2 either way.
#ifndef RMH_R2PAD
#define RMH_R2PAD 1
#endif
   static constexpr int R2S = D2 + ((!INPLACE_Y && NB > 1 && D2 % 16 == 0) ? RMH_R2PAD : 0);
   static constexpr int oR3 = 0, oR2 = INPLACE_Y ? 0 : NR * R3S, PF = INPLACE_Y ? NR * R3S : oR2 + NR * Q * R2S;
   // JS: plane stride of the intermediates of the back-transform (phase J) -- D^2 + D at p = 3, where 16 lines of an
   // element otherwise start in 4 LDS banks
#ifndef RMH_JPAD
#define RMH_JPAD ((P == 3) ? D : 0)
#endif
   static constexpr int JS = D2 + RMH_JPAD;
   static constexpr int oSA = 0, oM1 = D3, oR3c = oM1 + Q * S2, oSB = oR3c + Q2 * D, PCG = oSB + D * JS;
   // lo 4 (subcell residual distribution) extras behind the work region: the face rows -- GL-tested for the HO part, and
   // Bernstein-tested s rows for the RD solver (a second set when HO and RD share the kernel) -- whose slots hold the
   // sub-mesh node positions until the subcell pass has consumed them (it runs in front of the face rows); then the
   // subcell data [3][NS].  The lumped face flux per dof is written after the column pass (nodes, u, traces and U1 are dead
   // by then) behind the in-place test tensors, and read by the RD part in front of the PCG.
   static constexpr int NS = P * P * P;
   static constexpr int RF = 6 * Q * D; // face rows tested along q2
   // One-element workgroups of the lo 4 kernels (p >= 4) are short of LDS, not of lanes: there the lumped face flux lives behind
   // the face rows instead of in the work region, and the tables of the fused limiter (+108: stencil and box table, see phase
   // J; +2: the element's sum of the right-hand side and its volume, kept from the PCG prelude to the constant-mode completion,
   // see batch_dot_keep2) go where the GL-tested face rows were -- consumed in phase G, long before the PCG prelude.  p = 6:
   // 35 -> 32 KB per workgroup (room for a fifth workgroup per CU, which the registers do not admit: see WAVES_PER_SIMD);
   // p = 4, 5: +-0.
   static constexpr bool SLIM = LO4 && NB == 1;
   static constexpr int RFA = RF * (BOTH ? 2 : 1); // all face rows
   static constexpr int XT = LO4 ? cmax(RFA + (SLIM ? D3 : 0), 3 * D3) : RF;
   static constexpr int oDufW = cmax(cmax(PF, PCG), oU1); // (behind the traces, which the lumped fluxes read while they write it, and behind everything the RD part uses)
   static constexpr int W = SLIM ? cmax(PA, cmax(PF, PCG)) : cmax(PA, cmax(LO4 ? oDufW + D3 : PF, PCG + 110));
   static constexpr int oF = W;                       // s*jump rows (GL basis) -- or the s rows in the RD-only kernel
   static constexpr int oF2 = BOTH ? oF + RF : oF;    // s rows (Bernstein basis) of the RD solver
   static constexpr int oXs = oF;                     // sub-mesh nodes [3][D3], phases A-B
   static constexpr int oDuf = SLIM ? oF + RFA : oDufW;
   static constexpr int oLim = SLIM ? oF : PCG;       // limiter tables [108] and the two kept sums
   static constexpr int oKeep = oLim + 108;
   static_assert(!SLIM || RF >= 110, "no room for the limiter tables in the face rows");
   static constexpr int oSub = oF + XT;
   // element block stride: 16-byte aligned, and == 2 (mod 32) doubles so that the same offset of
   // neighbouring elements (two elements share most wavefronts) falls into different LDS banks
   static constexpr int EL0 = W + XT + (LO4 ? 3 * NS : 0);
   static constexpr int EL = (NB == 1) ? EL0 + (EL0 & 1) : EL0 + ((2 - EL0 % 32) + 32) % 32;
   // partial sums of the generic reductions: chunks of 64 dofs (one wavefront each), 8 chunks for small elements
   static constexpr int DOT_CH = D3 >= 64 ? (D3 + 63) / 64 : 8;
   static constexpr bool WAVE_ALIGNED = (D3 % 64) == 0; // every (round, wavefront) holds one element
   static constexpr int PART = (WAVE_ALIGNED && DR <= 2) ? 0 : cmax(8, 2 * DOT_CH) * NB; // (the DPP paths need none)
   // fused stage: the 27 stencil indices of every element ([NB][27] ints), parked in LDS from phase A to the PCG prelude
   static constexpr int STI = (NB * 27 + 1) / 2;
   // split columns (p = 6, see ho_kernel2 phase C): w detJ of the columns that three lanes share lives in LDS, [column][qz]
   static constexpr bool CSPL = NT == 128 && NB == 1 && Q2 > 64 && Q % 3 == 0 && 3 * (Q2 - 64) <= 64 && 6 * Q <= 64;
   static constexpr int WDL = CSPL ? (Q2 - 64) * Q : 0;
   static constexpr int LDS_DOUBLES = NB * EL + 4 * NB + 8 + T::N2S + PART + STI + WDL;
   // LDS allocation granule: a 54 096-byte kernel ran two workgroups per CU, a 52 560-byte one three (measured:
   // 6.3 k vs 8.6 k MDOFs*stage/s); 2 KiB granules are consistent with that
   static constexpr int LDS_BYTES = (8 * LDS_DOUBLES + 2047) / 2048 * 2048;
   // workgroups per CU the LDS budget admits (160 KiB); launch bounds ask for the matching registers
   static constexpr int WG_PER_CU = cmax(1, (160 * 1024) / LDS_BYTES > 1024 / NT ? 1024 / NT : (160 * 1024) / LDS_BYTES);
   // launch bound: wavefronts per SIMD that the LDS budget admits (a workgroup has NT / 64 wavefronts on 4 SIMDs)
   static constexpr int WAVES_PER_SIMD0 = cmax(1, WG_PER_CU * (NT / 64) / 4);
   // p = 6: LDS admits 5 workgroups of 2 wavefronts per CU = 2.5 per SIMD, which the registers only allow at <= 168
   // VGPRs.  Asking for 3 costs 100 B/lane of scratch in the column phase and still wins (9.65 k -> 10.0 k
   // MDOFs*stage/s; with the x-leg basis rows in registers through the PCG loop it was 248 B/lane and -12 %).
   // (the compiler honours the bound only as far as the kernel's LDS admits that occupancy)
#ifndef RMH_WAVES6
#define RMH_WAVES6 3
#endif
#ifndef RMH_WAVES5
#define RMH_WAVES5 3
#endif
   // (p = 4, round 4: four wavefronts per SIMD -- 128 VGPRs, 140 B/lane of scratch -- 20.8 k -> 17.4 k MDOFs*stage/s on cube01_hex -rs 5)
   // (lo 4 at p = 6, round 4: with the slim layout below the kernel's LDS admits a fifth workgroup per CU, i.e. 2.5 wavefronts per
   // SIMD; asking for 3 gives 168 VGPRs + 76 B/lane of scratch and 10.0 k instead of 11.7 k MDOFs*stage/s -- it stays at 2)
   // (lo 4 stage at p = 6, round 5: 183 VGPRs, and the LDS admits 5 workgroups = 2.5 wavefronts per SIMD.  A launch bound of 3 is
   // ignored as long as the compiler sees the 32 KB of static LDS -- it clamps the request to what the LDS admits, rounded down;
   // with the work region as dynamic LDS the bound is honoured (168 VGPRs, 68 B/lane of scratch, five resident workgroups per CU)
   // and the stage is 11 % SLOWER: 12.36 k -> 10.98 k MDOFs*stage/s.  It stays at 2, profiles/r05_lo4_split.txt.)
   // (p = 4 with the x-contracted node lines: 12.6 KB per workgroup; the 2 KiB model above says 11 workgroups per CU, the hardware runs 12 --
   // LDS is allocated in finer granules: measured, cube01_hex -rs 5 +2.5 % with the bound kept at three wavefronts per SIMD)
   static constexpr int WAVES_PER_SIMD = (P == 6 && !LO4) ? RMH_WAVES6 : ((P == 5 && !LO4) ? RMH_WAVES5 : ((P == 4 && !LO4) ? 3 : (WAVES_PER_SIMD0 > 8 ? 8 : WAVES_PER_SIMD0)));
};

// v + (v of the lane selected by the DPP control), lanes outside row_mask add 0
template <int CTRL, int ROW_MASK>
__device__ inline double dpp_add(double v)
{
   const int lo = __double2loint(v), hi = __double2hiint(v);
   const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, false);
   const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, false);
   return v + __hiloint2double(hi2, lo2);
}

// the value of the lane selected by the DPP control (0 where the control has no source lane)
template <int CTRL>
__device__ inline double dpp_value(double v)
{
   const int lo = __double2loint(v), hi = __double2hiint(v);
   // (bound_ctrl: a lane without a source gets 0 from the instruction itself -- with all rows and banks enabled the destination
   // then needs no zero-initialisation: one v_mov_b32 less per DPP move, 84 per split-column wavefront at p = 6)
   const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
   const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
   return __hiloint2double(hi2, lo2);
}

// the same for controls that give every lane a valid source (quad_perm, row mirrors over all rows): no
// "old" value, hence no zero-initialisation of the destination
template <int CTRL>
__device__ inline double dpp_add_all(double v)
{
#if defined(__HIP_DEVICE_COMPILE__)
   const int lo = __double2loint(v), hi = __double2hiint(v);
   const int lo2 = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, false);
   const int hi2 = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, false);
   return v + __hiloint2double(hi2, lo2);
#else
   return dpp_add<CTRL, 0xF>(v);
#endif
}

// v_permlane32_swap (gfx950): the upper 32 lanes of a are exchanged with the lower 32 lanes of b; afterwards
// a = {a[0..31], b[0..31]} and b = {a[32..63], b[32..63]}
__device__ inline void swap32(double &a, double &b)
{
#if defined(__HIP_DEVICE_COMPILE__)
   const auto r0 = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
   const auto r1 = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
   a = __hiloint2double((int)r1[0], (int)r0[0]);
   b = __hiloint2double((int)r1[1], (int)r0[1]);
#elif defined(HIPEMU)
   hipemu_permlane32_swap(a, b);
#endif
}

// v_permlane16_swap (gfx950): the odd 16-lane rows of a are exchanged with the even rows of b; afterwards
// a = {a.row0, b.row0, a.row2, b.row2} and b = {a.row1, b.row1, a.row3, b.row3}
__device__ inline void swap16(double &a, double &b)
{
#if defined(__HIP_DEVICE_COMPILE__)
   const auto r0 = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
   const auto r1 = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
   a = __hiloint2double((int)r1[0], (int)r0[0]);
   b = __hiloint2double((int)r1[1], (int)r0[1]);
#elif defined(HIPEMU)
   hipemu_permlane16_swap(a, b);
#endif
}

template <int CTRL, int ROW_MASK, bool IS_MIN>
__device__ inline double dpp_minmax(double v)
{
   const int lo = __double2loint(v), hi = __double2hiint(v);
   const int lo2 = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xF, false);
   const int hi2 = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xF, false);
   const double o = __hiloint2double(hi2, lo2);
   return IS_MIN ? fmin(v, o) : fmax(v, o);
}

// min or max over the 64 lanes of a wavefront; valid in lane 63
template <bool IS_MIN>
__device__ inline double wave_minmax(double v)
{
   v = dpp_minmax<0xB1, 0xF, IS_MIN>(v);
   v = dpp_minmax<0x4E, 0xF, IS_MIN>(v);
   v = dpp_minmax<0x141, 0xF, IS_MIN>(v);
   v = dpp_minmax<0x140, 0xF, IS_MIN>(v);
   v = dpp_minmax<0x142, 0xA, IS_MIN>(v);
   v = dpp_minmax<0x143, 0xC, IS_MIN>(v);
   return v;
}

// the value of lane L of the wavefront in all its lanes (on the device: two v_readlane_b32 into a scalar register pair)
template <int L>
__device__ inline double wave_bcast(double v)
{
#if defined(__HIP_DEVICE_COMPILE__)
   return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), L), __builtin_amdgcn_readlane(__double2loint(v), L));
#else
   return __shfl(v, L);
#endif
}

// Element sums where a wavefront holds whole elements in its dof rounds (p = 3: 64 dofs): the totals of the DPP reduction
// go from their lane straight to all lanes of the wavefront -- the lanes that want them -- instead of through LDS and a
// workgroup barrier (same sums, same order)
#ifndef RMH_FACE_MAJOR
#define RMH_FACE_MAJOR (P == 3)
#endif
// The Q2 mesh nodes (and node velocities) reach the stage kernel in the HIERARCHICAL form of the quadratic Lagrange basis on
// {0, 1/2, 1} along the directions of the bit mask RMH_HIER (1: x, 2: y, 4: z): (n0, n1 - n0, n2 - n0) -- a linear map,
// applied once on the host when the context is made (rmh_api.hip), so x0 + t v stays what the kernel forms.  Along such a
// direction the basis reads (1, L1, L2), its derivative (0, dL1, dL2) (the L sum to one, the dL to zero): a 1-D interpolation
// of the geometry costs two FMAs instead of three.  x and z: 153 of the 958 FP64 instructions of a p = 3 quadrature column
// (-16 % of the column pass, -7 % of the stage kernel's FP64 work; the same absolute saving per column and point at every
// order).  y as well would save 36 more, but that form of the y-leg makes the compiler spill at every order but 4 (p = 6: 44
// -> 484 B/lane of scratch, 20.7 k -> 12.7 k MDOFs*stage/s).  Same polynomial, evaluated from differences: results differ from
// the nodal evaluation by round-off.
// (RMH_HIER itself is defined in front of K2Cfg, whose layout depends on it)
#ifndef RMH_WAVE_DOT
#define RMH_WAVE_DOT 1
#endif

// sum over the 64 lanes of a wavefront in a fixed order; valid in lane 63
__device__ inline double wave_sum(double v)
{
   v = dpp_add_all<0xB1>(v);
   v = dpp_add_all<0x4E>(v);
   v = dpp_add_all<0x141>(v);
   v = dpp_add_all<0x140>(v);
   v = dpp_add<0x142, 0xA>(v);
   v = dpp_add<0x143, 0xC>(v);
   return v;
}

// Two wavefront sums for the price of one (one-element workgroups, round 6): the half-wave swap puts the lower halves of both values
// into lanes 0..31 and the upper halves into lanes 32..63, one addition folds the halves, and the five DPP steps of a 32-lane sum
// reduce both at once: the total of a is valid in lane 31, that of b in lane 63.  17 instead of 36 cross-lane / add instructions.
// (A fixed order, independent of where the element is -- but not wave_sum's: lane i and lane i + 32 meet first.)
// Measured (A/B in one process, round 6): p = 6 (two wavefronts per element) +0.5 %, lo 4 at p = 6 +0.55 %, p = 4 +-0,
// p = 5 -0.3 % -> on where an element spans more than one wavefront.
#ifndef RMH_PACKED_SUMS
#define RMH_PACKED_SUMS (NW > 1)
#endif
__device__ inline double wave_sum2(double a, double b)
{
   swap32(a, b);
   double x = a + b;
   x = dpp_add_all<0xB1>(x);
   x = dpp_add_all<0x4E>(x);
   x = dpp_add_all<0x141>(x);
   x = dpp_add_all<0x140>(x);
   x = dpp_add<0x142, 0xA>(x);
   return x;
}
// the same for a minimum and a maximum (carried as the minimum of the negated values): min(a) in lane 31, -max(b) in lane 63
__device__ inline double wave_min_negmax(double a, double b)
{
   double nb = -b;
   swap32(a, nb);
   double x = fmin(a, nb);
   x = dpp_minmax<0xB1, 0xF, true>(x);
   x = dpp_minmax<0x4E, 0xF, true>(x);
   x = dpp_minmax<0x141, 0xF, true>(x);
   x = dpp_minmax<0x140, 0xF, true>(x);
   x = dpp_minmax<0x142, 0xA, true>(x);
   return x;
}

// A fresh view of the constant table of order P.  Entries read through one view cannot be merged with (or
// hoisted next to) reads through another.  At p >= 5 the compiler otherwise keeps every table entry of the
// kernel live in scalar registers from its first use to its last and spills them to VGPR lanes (p = 6: more
// v_readlane than FMA instructions); views are taken per unrolled quadrature plane.  At p <= 4 the table fits
// the scalar registers and re-loading costs more than the few spills (measured: p = 3 -8 %), so the view is
// the table itself.
// Pointers into the constant table.  p >= 5 (table views, below): typed with their address space and made opaque as
// POINTERS -- an SGPR pair, entries are scalar loads at immediate offsets from it.  With an opaque index added to the symbol
// instead (rounds 1-2) every single entry got its own s_getpc_b64 + four 32-bit adds in front of its s_load: the y-leg of the
// test contractions at p = 6 had 1742 scalar instructions for 252 FMAs; with the typed pointer the p = 6 stage has 1583
// instead of 5057 scalar instructions per wavefront and runs 10 % faster (p = 5 +10 %, lo 4 at p = 6 +12 %, bit-identical).
// Views from p = 4 on (RMH_VIEW_MINP): with cheap views p = 4 gains 2.5-3 % over the plain symbol (cube01_hex -rs 5: 18.65 k ->
// 19.17 k; fewer scalar-register spills to VGPR lanes), p = 3 loses 1.3-2.7 %, p = 2 +-0; p <= 3 keep the plain symbol
// (typed but not opaque there: p = 4 -1.8 %, p = 3 +0.2 %).
#if defined(__HIP_DEVICE_COMPILE__)
typedef const double __attribute__((address_space(4))) *tabp_const;
#else
typedef const double *tabp_const;
#endif
template <int P>
using tabp_t = std::conditional_t<(P >= RMH_VIEW_MINP), tabp_const, const double *>;
template <int P>
__device__ inline tabp_t<P> tab_view()
{
   tabp_t<P> t = (tabp_t<P>)c_tab[P];
#if defined(__HIP_DEVICE_COMPILE__)
   if constexpr (P >= RMH_VIEW_MINP) { asm volatile("" : "+s"(t)); }
#endif
   return t;
}
// the same view typed in the constant address space for any order (p <= 3 experiments: phases whose tables do not fit the scalar
// registers beside the kernel's long-lived scalars -- y-leg: 3 x 24 doubles -- spill those to VGPR lanes, v_writelane / v_readlane)
template <int P>
__device__ inline tabp_const tab_view_c()
{
   tabp_const t = (tabp_const)c_tab[P];
#if defined(__HIP_DEVICE_COMPILE__)
   asm volatile("" : "+s"(t));
#endif
   return t;
}
#define RMH_TAB() tab_view<P>()
#define RMH_TABK() (P >= RMH_VIEW_MINP ? tab_view<P>() : gtb)

// The upwind face speeds come from a table made once per context (face_geom_kernel): w_q v.n_out(q, t) is a quadratic in the
// pseudo-time t (the mesh moves linearly), three coefficients per face quadrature point.  Rounds 2-3 kept the p = 3 HO / lo 5
// kernels on the node-based face rows (~315 FMAs per row): at the board's power limit the FP64 work saved and the HBM bytes
// added cancelled (tools/power_probe.py).  With the hierarchical nodes of the column pass (RMH_HIER) the table wins there too --
// +2.0 ... +2.5 % over 100 steps, +4.6 % at -rs 4, +5.2 % with the converged mass solve -- and the node-based path is gone
// (tools/experiments/r04_node_face_rows.patch).
// x-leg basis rows of a thread's dofs in registers through the PCG loop (see ho_kernel2, phase G)
#ifndef RMH_CBG_REG
#define RMH_CBG_REG (DR * Q <= 12)
#endif
// "Some element of the batch is still active": many work-items store the same 1 into one LDS word between two barriers.  A
// relaxed atomic store -- the same ds_write_b32 on the device -- says so, and the host emulation under ThreadSanitizer
// (tests/test_sanitizers.py) then reports real races only.
__device__ inline void raise_flag(int *flag)
{
#if defined(__HIP_DEVICE_COMPILE__)
   __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
   __atomic_store_n(flag, 1, __ATOMIC_RELAXED);
#endif
}

// An LDS offset (in doubles) the compiler must treat as a value it knows nothing about.  Why: an address "lane part +
// large constant" makes the DS load/store optimizer rebase EVERY pair of 8-byte reads it merges into a ds_read2_b64 (whose two
// offsets reach 2040 bytes) -- v_mov_b32 const, v_mad_u32_u24 lane, stride, const per pair: at p = 6 the column pass carried 158
// v_mad + 150 v_mov for its 98 U1 reads and its table rows (13 % of the phase's VALU instructions).  With the whole base made
// opaque the constants that remain are the small offsets inside a row, which fit the instruction.
__device__ inline int opaque_lds_offset(int x)
{
#if defined(__HIP_DEVICE_COMPILE__)
   asm volatile("" : "+v"(x));
#endif
   return x;
}

// nothing is scheduled across this point (keeps the LDS table reads of the split columns next to their uses: hoisted
// to the top of a quadrature-point loop they cost ~70 VGPRs)
__device__ inline void sched_fence()
{
#if defined(__HIP_DEVICE_COMPILE__)
   __builtin_amdgcn_sched_barrier(0);
#endif
}

// a / b for well-scaled operands (mass, dt, PCG scalars: no denormals, no overflow, b != 0): hardware reciprocal,
// two Newton steps and one correction of the quotient -- the core of the IEEE expansion without its scaling and
// special-case fix-up (8 instead of ~14 instructions; ~30 divisions per wavefront were 12 % of the stage kernel's
// VALU instructions).  The result is within 1 ulp of a / b, not always correctly rounded.
#ifndef RMH_FAST_DIV
#define RMH_FAST_DIV 1
#endif
__device__ inline double fdiv(double a, double b)
{
#if defined(__HIP_DEVICE_COMPILE__) && RMH_FAST_DIV
   double r = __builtin_amdgcn_rcp(b);
   r = fma(fma(-b, r, 1.0), r, r);
   r = fma(fma(-b, r, 1.0), r, r);
   const double q = a * r;
   return fma(fma(-b, q, a), r, q);
#else
   return a / b;
#endif
}

// The same quotient with the refined reciprocal of b made once (fdiv_rcp) and shared by several divisions by the same b --
// the limiter divides by dt twice per dof: a / b costs three instructions instead of eight, the value is fdiv's bit for bit
__device__ inline double fdiv_rcp(double b)
{
#if defined(__HIP_DEVICE_COMPILE__) && RMH_FAST_DIV
   double r = __builtin_amdgcn_rcp(b);
   r = fma(fma(-b, r, 1.0), r, r);
   return fma(fma(-b, r, 1.0), r, r);
#else
   return b;
#endif
}
__device__ inline double fdiv_by(double a, double b, double r)
{
#if defined(__HIP_DEVICE_COMPILE__) && RMH_FAST_DIV
   const double q = a * r;
   return fma(fma(-b, q, a), r, q);
#else
   (void)r;
   return a / b;
#endif
}

// The kernel's argument struct re-read from the kernarg segment at the point of use (see ho_kernel2, phase I).  The
// view is typed in the constant address space so that its fields arrive by scalar loads (through a generic pointer
// they were vector loads from global memory, each followed by a full s_waitcnt vmcnt(0) in front of its first use).
#if defined(__HIP_DEVICE_COMPILE__)
typedef const HoArgs __attribute__((address_space(4))) &LateArgs;
__device__ inline LateArgs late_args(const HoArgs &)
{
   int z = 0;
   asm volatile("" : "+s"(z));
   typedef const char __attribute__((address_space(4))) *kptr;
   return *(const HoArgs __attribute__((address_space(4))) *)((kptr)__builtin_amdgcn_kernarg_segment_ptr() + z);
}
#else
typedef const HoArgs &LateArgs;
__device__ inline LateArgs late_args(const HoArgs &a) { return a; }
#endif

// Sum over the dofs of each element of the batch: values v[r] of the dof role -> out[r] (the
// element total, broadcast back to the dof threads).  ONE barrier per call: results go through a
// ring of three LDS buffers (s_acc3[3][NB]); the buffer of the call before the previous one is
// re-zeroed here, when every thread is provably past its reads.
//   fast path (p = 3: D3 = 64 dofs = one wavefront per element and round, two rounds): the two
//   rounds are reduced together with a halving butterfly -- 6 cross-lane steps for both values;
//   generic path: LDS float64 atomics.
template <class C>
__device__ inline void batch_dot(const int tid, const double (&v)[C::DR], double (&out)[C::DR], double *lds, double *s_acc3, int &ring)
{
   double *cur = s_acc3 + ring * C::NB;
   double *old = s_acc3 + ((ring + 1) % 4) * C::NB; // (a slot is reused two calls later at the earliest)
   if (C::WAVE_ALIGNED && C::DR == 2)
   {
      const int lane = tid & 63, wave = tid >> 6;
      double v0 = v[0];
      double v1 = (tid + C::NT < C::NB * C::D3) ? v[C::DR == 2 ? 1 : 0] : 0.0;
      // lanes 0..31 gather round 0, lanes 32..63 round 1 (one half-wave swap); then DPP row reductions
      // (VALU only, no LDS)
      swap32(v0, v1);
      double x = v0 + v1;
      x = dpp_add_all<0xB1>(x);  // quad_perm [1,0,3,2]
      x = dpp_add_all<0x4E>(x);  // quad_perm [2,3,0,1]
      x = dpp_add_all<0x141>(x); // row_half_mirror
      x = dpp_add_all<0x140>(x); // row_mirror: every lane of a row holds the row total
      x = dpp_add<0x142, 0xA>(x); // row_bcast:15 into rows 1 and 3: half-wave totals
      if (RMH_WAVE_DOT)
      {
         out[0] = wave_bcast<31>(x);
         out[C::DR == 2 ? 1 : 0] = wave_bcast<63>(x);
         return;
      }
      if (lane == 31) { cur[wave] = x; }
      if (lane == 63 && (C::NT / C::D3 + wave) < C::NB) { cur[C::NT / C::D3 + wave] = x; }
   }
   else if (C::WAVE_ALIGNED && C::DR == 1)
   {
      // one round: every wavefront holds one element
      const double x = wave_sum((tid < C::NB * C::D3) ? v[0] : 0.0);
      if ((tid & 63) == 63 && (tid >> 6) < C::NB) { cur[tid >> 6] = x; }
   }
   else if (C::NB == 1)
   {
      // one element per workgroup (p = 6): every thread adds its rounds, every wavefront reduces by DPP, the
      // wavefront totals are added in order -- one barrier and no round trip of the values through LDS (the order of
      // the sum is fixed by the dof -> (thread, round) map, which does not depend on where the element is)
      constexpr int NW = C::NT / 64;
      static_assert(C::NB != 1 || 4 * NW <= C::PART, "partial-sum ring does not fit");
      double *slot = s_acc3 + 4 * C::NB + 8 + C::N2S + ring * NW;
      double x = 0.0;
#pragma unroll
      for (int r = 0; r < C::DR; r++) { x += (tid + r * C::NT < C::D3) ? v[r] : 0.0; }
      x = wave_sum(x);
      if (NW == 1)
      {
         // (p = 4, 5: the workgroup IS the wavefront -- the total goes from lane 63 to all lanes through two v_readlane, not
         // through an LDS word and a barrier: the same bits, ~150 cycles less per reduction, eight reductions per stage)
         const double tot1 = wave_bcast<63>(x);
#pragma unroll
         for (int r = 0; r < C::DR; r++) { out[r] = (tid + r * C::NT < C::D3) ? tot1 : 0.0; }
         return;
      }
      if ((tid & 63) == 63) { slot[tid >> 6] = x; }
      __syncthreads();
      double tot = slot[0];
#pragma unroll
      for (int w = 1; w < NW; w++) { tot += slot[w]; }
#pragma unroll
      for (int r = 0; r < C::DR; r++) { out[r] = (tid + r * C::NT < C::D3) ? tot : 0.0; }
      ring = (ring + 1) % 4;
      return;
   }
   else
   {
      // generic orders: deterministic two-level sum (no atomics: the result must not depend on the
      // arrival order, nor on the slot of the element in the batch -- rank-count invariance of the whole
      // run is checked bit for bit).  Values go to the sB slot of the element block (free outside phase J);
      // every 64-dof chunk of an element is summed by one wavefront (DPP row reductions, fixed order),
      // the chunk sums of an element are added in order.
      (void)old;
      (void)cur;
#pragma unroll
      for (int r = 0; r < C::DR; r++)
      {
         const int t = tid + r * C::NT;
         if (t < C::NB * C::D3) { (lds + (t / C::D3) * C::EL)[C::oSB + t % C::D3] = v[r]; }
      }
      __syncthreads();
      double *part = s_acc3 + 4 * C::NB + 8 + C::N2S; // [NB][CH], behind the table copy
      constexpr int CH = C::DOT_CH;
      if (C::D3 >= 64)
      {
         // CH chunks of 64 dofs per element, one wavefront each
         const int lane = tid & 63, wave = tid >> 6;
         for (int k = wave; k < C::NB * CH; k += C::NT / 64)
         {
            const int i = (k % CH) * 64 + lane;
            double x = (i < C::D3) ? (lds + (k / CH) * C::EL)[C::oSB + i] : 0.0;
            x = wave_sum(x);
            if (lane == 63) { part[k] = x; }
         }
      }
      else
      {
         // small elements (p = 1, 2): CH chunks per element are summed serially by one thread each
         constexpr int CL = (C::D3 + CH - 1) / CH;
         for (int k = tid; k < C::NB * CH; k += C::NT)
         {
            const double *src = lds + (k / CH) * C::EL + C::oSB;
            const int i0 = (k % CH) * CL;
            double acc = 0.0;
            for (int i = i0; i < i0 + CL && i < C::D3; i++) { acc += src[i]; }
            part[k] = acc;
         }
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < C::DR; r++)
      {
         const int t = tid + r * C::NT;
         double acc = 0.0;
         if (t < C::NB * C::D3)
         {
            const double *pp = part + (t / C::D3) * CH;
#pragma unroll
            for (int c = 0; c < CH; c++) { acc += pp[c]; }
         }
         out[r] = acc;
      }
      ring = (ring + 1) % 4;
      return;
   }
   __syncthreads();
#pragma unroll
   for (int r = 0; r < C::DR; r++)
   {
      const int t = tid + r * C::NT;
      out[r] = (t < C::NB * C::D3) ? cur[t / C::D3] : 0.0;
   }
   ring = (ring + 1) % 4;
}

// Two element sums at once (v -> outv, w -> outw): for p = 3 the two half-wave results share one row reduction --
// after the half-wave swap each value occupies two 16-lane rows per round; a row swap between the two values puts
// {v round 0, w round 0, v round 1, w round 1} into the four rows, which the four DPP row steps then reduce together.
// One barrier instead of two.  Other orders: two calls of batch_dot.
template <class C>
__device__ inline void batch_dot2(const int tid, const double (&v)[C::DR], const double (&w)[C::DR], double (&outv)[C::DR],
                                  double (&outw)[C::DR], double *lds, double *s_acc3, int &ring)
{
   if (C::WAVE_ALIGNED && C::DR == 2)
   {
      const int lane = tid & 63, wave = tid >> 6;
      double *curv = s_acc3 + ring * C::NB, *curw = s_acc3 + ((ring + 1) % 4) * C::NB;
      const bool has1 = tid + C::NT < C::NB * C::D3;
      double v0 = v[0], v1 = has1 ? v[C::DR == 2 ? 1 : 0] : 0.0;
      double w0 = w[0], w1 = has1 ? w[C::DR == 2 ? 1 : 0] : 0.0;
      swap32(v0, v1);
      swap32(w0, w1);
      double xa = v0 + v1, xb = w0 + w1;
      swap16(xa, xb);
      double x = xa + xb;
      x = dpp_add_all<0xB1>(x);
      x = dpp_add_all<0x4E>(x);
      x = dpp_add_all<0x141>(x);
      x = dpp_add_all<0x140>(x);
      if (RMH_WAVE_DOT)
      {
         outv[0] = wave_bcast<15>(x);
         outw[0] = wave_bcast<31>(x);
         outv[C::DR == 2 ? 1 : 0] = wave_bcast<47>(x);
         outw[C::DR == 2 ? 1 : 0] = wave_bcast<63>(x);
         return;
      }
      const int e1 = C::NT / C::D3 + wave;
      if (lane == 15) { curv[wave] = x; }
      if (lane == 31) { curw[wave] = x; }
      if (lane == 47 && e1 < C::NB) { curv[e1] = x; }
      if (lane == 63 && e1 < C::NB) { curw[e1] = x; }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < C::DR; r++)
      {
         const int t = tid + r * C::NT;
         outv[r] = (t < C::NB * C::D3) ? curv[t / C::D3] : 0.0;
         outw[r] = (t < C::NB * C::D3) ? curw[t / C::D3] : 0.0;
      }
      ring = (ring + 2) % 4;
   }
   else if (C::NB == 1)
   {
      // (see batch_dot: both sums behind one barrier)
      constexpr int NW = C::NT / 64;
      double *slotv = s_acc3 + 4 * C::NB + 8 + C::N2S + ring * NW;
      double *slotw = s_acc3 + 4 * C::NB + 8 + C::N2S + ((ring + 1) % 4) * NW;
      double x = 0.0, y = 0.0;
#pragma unroll
      for (int r = 0; r < C::DR; r++)
      {
         const bool in = tid + r * C::NT < C::D3;
         x += in ? v[r] : 0.0;
         y += in ? w[r] : 0.0;
      }
      if (RMH_PACKED_SUMS)
      {
         x = wave_sum2(x, y); // (total of v in lane 31, of w in lane 63)
         y = x;
      }
      else
      {
         x = wave_sum(x);
         y = wave_sum(y);
      }
      if (NW == 1)
      {
         const double tv1 = wave_bcast<(RMH_PACKED_SUMS ? 31 : 63)>(x), tw1 = wave_bcast<63>(y); // (one wavefront: see batch_dot)
#pragma unroll
         for (int r = 0; r < C::DR; r++)
         {
            const bool in = tid + r * C::NT < C::D3;
            outv[r] = in ? tv1 : 0.0;
            outw[r] = in ? tw1 : 0.0;
         }
         return;
      }
      if (RMH_PACKED_SUMS)
      {
         if ((tid & 63) == 31) { slotv[tid >> 6] = x; }
         if ((tid & 63) == 63) { slotw[tid >> 6] = y; }
      }
      else if ((tid & 63) == 63) { slotv[tid >> 6] = x; slotw[tid >> 6] = y; }
      __syncthreads();
      double totv = slotv[0], totw = slotw[0];
#pragma unroll
      for (int k = 1; k < NW; k++) { totv += slotv[k]; totw += slotw[k]; }
#pragma unroll
      for (int r = 0; r < C::DR; r++)
      {
         const bool in = tid + r * C::NT < C::D3;
         outv[r] = in ? totv : 0.0;
         outw[r] = in ? totw : 0.0;
      }
      ring = (ring + 2) % 4;
   }
   else
   {
      batch_dot<C>(tid, v, outv, lds, s_acc3, ring);
      batch_dot<C>(tid, w, outw, lds, s_acc3, ring);
   }
}

// Three element sums behind one barrier, for the prelude of the mass solve: v -> outv (broadcast to the dof threads
// like batch_dot), and w, z -> the two KEPT doubles of the element block (oKeep, oKeep + 1), which later phases read
// from LDS (no registers held through the PCG loop).  The next reduction must be separated from this call by a barrier
// (three ring slots are in use in the one-element path): the PCG loop starts with one.
template <class C>
__device__ inline void batch_dot_keep2(const int tid, const double (&v)[C::DR], const double (&w)[C::DR], const double (&z)[C::DR],
                                       double (&outv)[C::DR], double *lds, double *s_acc3, int &ring)
{
   if (C::WAVE_ALIGNED && C::DR == 2)
   {
      const int lane = tid & 63, wave = tid >> 6;
      double *cur = s_acc3 + ring * C::NB;
      const bool has1 = tid + C::NT < C::NB * C::D3;
      double v0 = v[0], v1 = has1 ? v[C::DR == 2 ? 1 : 0] : 0.0;
      double w0 = w[0], w1 = has1 ? w[C::DR == 2 ? 1 : 0] : 0.0;
      double z0 = z[0], z1 = has1 ? z[C::DR == 2 ? 1 : 0] : 0.0;
      swap32(v0, v1);
      swap32(w0, w1);
      swap32(z0, z1);
      double x = v0 + v1;
      double xa = w0 + w1, xb = z0 + z1;
      swap16(xa, xb);
      double y = xa + xb; // rows: {w round 0, z round 0, w round 1, z round 1}
      x = dpp_add_all<0xB1>(x);
      y = dpp_add_all<0xB1>(y);
      x = dpp_add_all<0x4E>(x);
      y = dpp_add_all<0x4E>(y);
      x = dpp_add_all<0x141>(x);
      y = dpp_add_all<0x141>(y);
      x = dpp_add_all<0x140>(x);
      y = dpp_add_all<0x140>(y);
      x = dpp_add<0x142, 0xA>(x);
      const int e1 = C::NT / C::D3 + wave;
      if (RMH_WAVE_DOT)
      {
         // (the kept sums are read by the dof threads of their element -- this wavefront -- behind later barriers)
         if (lane == 15) { (lds + wave * C::EL)[C::oKeep] = y; }
         if (lane == 31) { (lds + wave * C::EL)[C::oKeep + 1] = y; }
         if (lane == 47 && e1 < C::NB) { (lds + e1 * C::EL)[C::oKeep] = y; }
         if (lane == 63 && e1 < C::NB) { (lds + e1 * C::EL)[C::oKeep + 1] = y; }
         outv[0] = wave_bcast<31>(x);
         outv[C::DR == 2 ? 1 : 0] = wave_bcast<63>(x);
         return;
      }
      if (lane == 31) { cur[wave] = x; }
      if (lane == 63 && e1 < C::NB) { cur[e1] = x; }
      if (lane == 15) { (lds + wave * C::EL)[C::oKeep] = y; }
      if (lane == 31) { (lds + wave * C::EL)[C::oKeep + 1] = y; }
      if (lane == 47 && e1 < C::NB) { (lds + e1 * C::EL)[C::oKeep] = y; }
      if (lane == 63 && e1 < C::NB) { (lds + e1 * C::EL)[C::oKeep + 1] = y; }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < C::DR; r++)
      {
         const int t = tid + r * C::NT;
         outv[r] = (t < C::NB * C::D3) ? cur[t / C::D3] : 0.0;
      }
      ring = (ring + 1) % 4;
   }
   else if (C::NB == 1)
   {
      constexpr int NW = C::NT / 64;
      double *slot = s_acc3 + 4 * C::NB + 8 + C::N2S;
      double *sv = slot + ring * NW, *sw = slot + ((ring + 1) % 4) * NW, *sz = slot + ((ring + 2) % 4) * NW;
      double x = 0.0, y = 0.0, q = 0.0;
#pragma unroll
      for (int r = 0; r < C::DR; r++)
      {
         const bool in = tid + r * C::NT < C::D3;
         x += in ? v[r] : 0.0;
         y += in ? w[r] : 0.0;
         q += in ? z[r] : 0.0;
      }
      if (RMH_PACKED_SUMS)
      {
         y = wave_sum2(y, q); // (the two kept sums share a reduction: w in lane 31, z in lane 63)
         q = y;
         x = wave_sum(x);
      }
      else
      {
         x = wave_sum(x);
         y = wave_sum(y);
         q = wave_sum(q);
      }
      if (NW == 1)
      {
         // (one wavefront: see batch_dot; lane 63 holds the totals and parks the two kept ones -- their readers are behind barriers)
         if (RMH_PACKED_SUMS)
         {
            if (tid == 31) { lds[C::oKeep] = y; }
            if (tid == 63) { lds[C::oKeep + 1] = q; }
         }
         else if (tid == 63) { lds[C::oKeep] = y; lds[C::oKeep + 1] = q; }
         const double tv1 = wave_bcast<63>(x);
#pragma unroll
         for (int r = 0; r < C::DR; r++) { outv[r] = (tid + r * C::NT < C::D3) ? tv1 : 0.0; }
         return;
      }
      if (RMH_PACKED_SUMS)
      {
         if ((tid & 63) == 31) { sw[tid >> 6] = y; }
         if ((tid & 63) == 63) { sv[tid >> 6] = x; sz[tid >> 6] = q; }
      }
      else if ((tid & 63) == 63) { sv[tid >> 6] = x; sw[tid >> 6] = y; sz[tid >> 6] = q; }
      __syncthreads();
      double totv = sv[0];
#pragma unroll
      for (int k = 1; k < NW; k++) { totv += sv[k]; }
      if (tid == 0)
      {
         double totw = sw[0], totz = sz[0];
#pragma unroll
         for (int k = 1; k < NW; k++) { totw += sw[k]; totz += sz[k]; }
         lds[C::oKeep] = totw;
         lds[C::oKeep + 1] = totz;
      }
#pragma unroll
      for (int r = 0; r < C::DR; r++) { outv[r] = (tid + r * C::NT < C::D3) ? totv : 0.0; }
      ring = (ring + 3) % 4;
   }
   else
   {
      double ow[C::DR], oz[C::DR];
      batch_dot<C>(tid, v, outv, lds, s_acc3, ring);
      batch_dot<C>(tid, w, ow, lds, s_acc3, ring);
      batch_dot<C>(tid, z, oz, lds, s_acc3, ring);
#pragma unroll
      for (int r = 0; r < C::DR; r++)
      {
         const int t = tid + r * C::NT;
         if (t < C::NB * C::D3 && t % C::D3 == 0)
         {
            (lds + (t / C::D3) * C::EL)[C::oKeep] = ow[r];
            (lds + (t / C::D3) * C::EL)[C::oKeep + 1] = oz[r];
         }
      }
   }
}

#include "rmh_diag.hpp" // RMH_STAMP / RMH_STAMP_FLUSH: empty unless the diagnostic build -DRMH_STAMPS

// Two-wavefront workgroups with one element (p = 6): the pencil-type phases have fewer tasks than a wavefront has
// lanes (49 x-pencils, 63 (q, iz) lines), so the second wavefront would only wait at the next barrier.  Instead both
// wavefronts take every task and each computes HALF of its outputs: the instruction stream of the phase halves.
// split_outputs<SPL, N>(wv, f) calls f(lo, hi) with compile-time bounds: [0, N) without splitting, [0, H) on wavefront 0
// and [H, N) on wavefront 1 otherwise (wv is wavefront-uniform: a scalar branch).
template <int V>
struct IntC
{
   static constexpr int value = V;
   constexpr operator int() const { return V; }
};
template <bool SPL, int N, class F>
__device__ inline void split_outputs(const int wv, F &&f)
{
   if constexpr (!SPL) { f(IntC<0>{}, IntC<N>{}); }
   else
   {
      constexpr int H = (N + 1) / 2;
      if (wv == 0) { f(IntC<0>{}, IntC<H>{}); }
      else { f(IntC<H>{}, IntC<N>{}); }
   }
}

// Lane-dependent axis arithmetic without select chains: n^c for an axis c in {0, 1, 2} (n^2 < 256) is a byte of one
// packed constant, (c + 1) % 3 and (c + 2) % 3 two bits of another (one shift-and-mask each; the ternaries they
// replace were a third of the integer instructions of phase A).
template <int N>
__device__ inline int axis_stride(const int c)
{
   static_assert(N * N < 256, "stride does not fit a byte");
   return ((1u | (unsigned)N << 8 | (unsigned)(N * N) << 16) >> (8 * c)) & 0xffu;
}
__device__ inline int axis_next(const int c) { return (0x09u >> (2 * c)) & 3u; }  // (c + 1) % 3
__device__ inline int axis_next2(const int c) { return (0x12u >> (2 * c)) & 3u; } // (c + 2) % 3

// Face geometry table.  On face f of an element, at the face quadrature point (q1, q2),
//    w_q1 w_q2 (v . n_out)(t),   n_out = +-(dX/dxi1 x dX/dxi2),  X(t) = x0 + t v,
// is the quadratic c0 + c1 t + c2 t^2 with c_k = +-w v.n_k, n_0 = T1x x T2x, n_1 = T1x x T2v + T1v x T2x, n_2 = T1v x T2v
// (T1x, T1v: tangents of x0 and of v); a static mesh (transport) has c1 = c2 = 0.  What the face rows of ho_kernel2
// evaluated from the 27 nodes every stage (DGTraceIntegrator set-up, remhos.cpp:651-657) is read from a table
// instead: 2 FMAs per point in place of ~50, for HBM bytes the stage has to spare.
// Round 5: one block of the table per FACE, not per element side (the reference's face quadrature data is per face too,
// remhos_lo.cpp:513-610).  The two sides of an interior face see the same nine nodes and opposite normals, so the block of
// the side with the low local face (side 0: slot 3 e + c) serves the neighbour's high face with the sign flipped -- the same
// bits, negated.  A high face keeps a block of its own ("orphan" slot, behind the 3 ne regular ones) where that does not
// hold: boundary faces, ghost neighbours, and periodic seams of a remap run, whose two node copies slide against each other
// (SURVEY 8d).  face_rows[e][6] (made on the host by rmh_create from the node data itself, bit by bit) holds the slot of
// every element face, bit 31 set = read with the sign flipped:
//    fgeo[slot][q2][k][q1]    (3 Q * Q doubles per slot; ~3.1 instead of 6 slots per element on a periodic lattice)
template <int P>
struct FaceGeo
{
   static constexpr int Q = K2Cfg<P>::Q, R = 6 * Q, SLOT = 3 * Q * Q;
};
constexpr int RMH_FACE_FLIP = (int)0x80000000u;

template <int P>
__global__ void face_geom_kernel(const double *x0, const double *vel, const double *tab, const int move, const int *face_rows, double *fgeo)
{
   using C = K2Cfg<P>;
   constexpr int Q = C::Q, R = 6 * Q;
   const size_t e = blockIdx.x;
   const double *X = x0 + e * 81, *V = vel + e * 81;
   for (int pt = threadIdx.x; pt < R * Q; pt += blockDim.x)
   {
      const int q2 = pt / R, r = pt % R;
      const int f = r / Q, q1 = r % Q;
      const int row = face_rows[e * 6 + f];
      if (row < 0) { continue; } // served by the neighbour's block
      const int c = f >> 1, side = f & 1;
      const int n0 = side ? 2 * axis_stride<3>(c) : 0, n1 = axis_stride<3>(axis_next(c)), n2 = axis_stride<3>(axis_next2(c));
      double t1x[3], t1v[3], t2x[3], t2v[3], vf[3];
      for (int comp = 0; comp < 3; comp++)
      {
         double s1x = 0, s1v = 0, s2x = 0, s2v = 0, sv = 0;
         for (int a2 = 0; a2 < 3; a2++)
         {
            const double L2 = tab[C::oL + q2 * 3 + a2], dL2 = tab[C::odL + q2 * 3 + a2];
            for (int a1 = 0; a1 < 3; a1++)
            {
               const double L1 = tab[C::oL + q1 * 3 + a1], dL1 = tab[C::odL + q1 * 3 + a1];
               const double x = X[comp * 27 + n0 + a1 * n1 + a2 * n2], v = V[comp * 27 + n0 + a1 * n1 + a2 * n2];
               s1x += dL1 * L2 * x;
               s2x += L1 * dL2 * x;
               s1v += dL1 * L2 * v;
               s2v += L1 * dL2 * v;
               sv += L1 * L2 * v;
            }
         }
         t1x[comp] = s1x; t2x[comp] = s2x; vf[comp] = sv;
         t1v[comp] = move ? s1v : 0.0;
         t2v[comp] = move ? s2v : 0.0;
      }
      double ck[3] = {0, 0, 0};
      for (int comp = 0; comp < 3; comp++)
      {
         const int i = (comp + 1) % 3, j = (comp + 2) % 3;
         ck[0] += vf[comp] * (t1x[i] * t2x[j] - t1x[j] * t2x[i]);
         ck[1] += vf[comp] * (t1x[i] * t2v[j] - t1x[j] * t2v[i] + t1v[i] * t2x[j] - t1v[j] * t2x[i]);
         ck[2] += vf[comp] * (t1v[i] * t2v[j] - t1v[j] * t2v[i]);
      }
      const double w = (side ? 1.0 : -1.0) * tab[C::oW + q1] * tab[C::oW + q2];
      for (int k = 0; k < 3; k++) { fgeo[(size_t)row * FaceGeo<P>::SLOT + (size_t)(q2 * 3 + k) * Q + q1] = w * ck[k]; }
   }
}

// Primary global loads of one element batch (phase A): face-neighbour indices, stencil indices (fused stage), Q2
// nodes of x0 and v, u.
template <class C, bool FUSED, bool ULN, int NLN, int NLS, int NLX, int NLU>
__device__ inline void load_batch(const HoArgs &a, const int e0, const int tid, int (&nbi)[NLN], int (&sti)[NLS], double (&gx0)[NLX],
                                  double (&gv)[NLX], double (&gu)[NLU])
{
   constexpr int NT = C::NT, NB = C::NB, D2 = C::D2, D3 = C::D3;
   // Straight-line loads: lanes past the end of a list load its last entry again (never stored) instead of branching
   // around the load.  With branches the compiler cannot count the loads in flight and drains ALL of them
   // (s_waitcnt vmcnt(0)) where only the neighbour indices -- issued first -- are needed to issue the trace loads.
#pragma unroll
   for (int j = 0; j < NLN; j++)
   {
      const int k = min(tid + j * NT, NB * 6 * D2 - 1);
      const int eb = k / (6 * D2), f = (k % (6 * D2)) / D2;
      nbi[j] = a.face_nbr[(size_t)min(e0 + eb, a.e_end - 1) * 6 + f];
   }
#pragma unroll
   for (int j = 0; j < NLS; j++)
   {
      const int k = min(tid + j * NT, NB * 27 - 1);
      sti[j] = -1;
      if (FUSED) { sti[j] = a.stencil27[(size_t)min(e0 + k / 27, a.e_end - 1) * 27 + k % 27]; }
   }
   if constexpr (C::XPK != 0)
   {
      // (node LINES: the three nodes of (comp, ay, az) go to one thread, which contracts them along x before they reach LDS)
#pragma unroll
      for (int j = 0; j < NLX / 3; j++)
      {
         const int k = min(tid + j * NT, NB * 27 - 1);
         const size_t base = (size_t)min(e0 + k / 27, a.e_end - 1) * 81 + 3 * (k % 27);
#pragma unroll
         for (int i = 0; i < 3; i++)
         {
            gx0[3 * j + i] = a.x0[base + i];
            gv[3 * j + i] = a.vel[base + i];
         }
      }
   }
   else
   {
#pragma unroll
      for (int j = 0; j < NLX; j++)
      {
         const int k = min(tid + j * NT, NB * 81 - 1);
         const int e = min(e0 + k / 81, a.e_end - 1);
         gx0[j] = a.x0[(size_t)e * 81 + k % 81];
         gv[j] = a.vel[(size_t)e * 81 + k % 81];
      }
   }
   if constexpr (ULN)
   {
      // (u LINES: the D values of a line (iy, iz) go to the thread that contracts them along x on their way to LDS, ho_kernel2 phase A;
      // tasks and threads as in the pencil phases -- split workgroups: both wavefronts load the line, each forms half the outputs)
      constexpr bool SPL = NT == 128 && NB == 1;
      constexpr int PNT = SPL ? 64 : NT, D = C::D;
      const int ptid = SPL ? (tid & 63) : tid;
#pragma unroll
      for (int j = 0; j < NLU / D; j++)
      {
         const int k = min(ptid + j * PNT, NB * D2 - 1);
         const double *line = a.u + (size_t)min(e0 + k / D2, a.e_end - 1) * D3 + D * (k % D2);
#pragma unroll
         for (int ix = 0; ix < D; ix++) { gu[j * D + ix] = line[ix]; }
      }
   }
   else
   {
#pragma unroll
      for (int j = 0; j < NLU; j++)
      {
         const int k = min(tid + j * NT, NB * D3 - 1);
         const int e = min(e0 + k / D3, a.e_end - 1);
         gu[j] = a.u[(size_t)e * D3 + k % D3];
      }
   }
}

// FUSED = false: HOSolver::CalcHOSolution (writes du_HO, lumped mass, element extrema of u).
// FUSED = true : the whole RK stage for -ho 3 -lo 5 -fct 2 (AdvectionOperator::Mult, remhos.cpp:1596-1916,
//                plus the RK3 vector update): HO as above, then MassBasedAvg (remhos_lo.cpp:247-324),
//                overlap bounds from the 27-element stencil (remhos_tools.cpp:432-495), ClipScale
//                (remhos_fct.cpp:449-541), y_out = a*x_base + b*(u + dt_rk*du), and the element extrema of
//                y_out for the next stage.  du_HO, du_LO, lumped mass and per-dof bounds never leave the CU.
// MODE 2        : PAResidualDistributionSubcell::CalcLOSolution (remhos_lo.cpp:1620-1802) with the same batching:
//                 z = K_vol u, lumped upwind face fluxes, sub-mesh motion and subcell fluctuations, nodal
//                 weights, du_LO; also writes the lumped mass and the element extrema (like the reference's
//                 RD solver does, remhos_lo.cpp:1702-1716).  No mass solve.
// the layout of ho_kernel2<P, MODE>.  The u slot is dropped where the room buys something: p = 4, whose one-wavefront workgroups then hold all
// three x-contractions of the node lines (XPK = 3 instead of 2: 23.5 -> 24.1 k instead of 23.9 k on cube01_hex -rs 5); p = 3, 5 +-0 (nothing to
// put there), p = 6 -0.3 % (the offsets behind the slot change parity).
#ifndef RMH_NOU
#define RMH_NOU (P == 4)
#endif
template <int P, int MODE>
using K2For = K2Cfg<P, (MODE >= 2), (MODE == 3), (MODE == 1 && (RMH_ULN) && (RMH_NOU))>;

template <int P, int MODE>
// MODE 3        : the whole RK stage for -ho 3 -lo 4 -fct 2: MODE 0 + MODE 2 + overlap bounds + ClipScale + RK
//                 update in one kernel (geometry, face data and u-contractions shared by HO and RD).
__global__ void __launch_bounds__((K2For<P, MODE>::NT), (K2For<P, MODE>::WAVES_PER_SIMD)) ho_kernel2(HoArgs a)
{
   constexpr bool FUSED = MODE == 1 || MODE == 3; // limiter + RK update at the end
   constexpr bool LO4 = MODE >= 2;                // subcell residual distribution pieces
   constexpr bool BOTH = MODE == 3;               // HO and RD in the same kernel
   constexpr bool HAS_HO = MODE != 2;
#ifdef RMH_STAMPS
   __shared__ unsigned long long s_stamp[32];
   if (threadIdx.x < 32) { s_stamp[threadIdx.x] = 0; }
   unsigned long long stamp_prev_ = clock64();
#endif
   using C = K2For<P, MODE>;
   constexpr int D = C::D, Q = C::Q, D2 = C::D2, D3 = C::D3, Q2 = C::Q2, NT = C::NT, NB = C::NB, DR = C::DR;
   constexpr int S2 = C::S2;
   constexpr int oXV = C::oXV, oU = C::oU, oNb = C::oNb, oU1 = C::oU1, oR3 = C::oR3, oR2 = C::oR2, oSA = C::oSA,
                 oM1 = C::oM1, oR3c = C::oR3c, oSB = C::oSB, oF = C::oF;
   __shared__ double lds[C::LDS_DOUBLES];
   // element-major blocks: element eb owns lds[eb*EL .. (eb+1)*EL)
#define RMH_W(eb) (lds + (eb) * C::EL)
   double *s_acc = lds + NB * C::EL;   // [3][NB] ring of reduction buffers (+ NB spare)
   int *s_flag = (int *)(s_acc + 4 * NB); // [4] "any element still active" flags (ring of 2 used)
   double *stab = s_acc + 4 * NB + 8;  // table copy for lane-dependent indexing
   int *s_sti = (int *)(stab + C::N2S + C::PART); // [NB][27] stencil indices (fused stage)
   double *s_wdl = stab + C::N2S + C::PART + C::STI; // [Q2 - 64][Q] w detJ of the split columns (p = 6)

   const int tid0 = threadIdx.x;
   static_assert(C::N3 <= RMH_TAB_STRIDE, "constant table too small");
   constexpr int oB = C::oB, oG = C::oG, oL = C::oL, odL = C::odL, oW = C::oW, oBg = C::oBg, oBg2 = C::oBg2,
                 oCi = C::oCi;

   // ---- phase A: loads ----------------------------------------------------------------------
   // (one batch of NB elements per workgroup)
   constexpr int NLX = C::XPK != 0 ? 3 * ((NB * 27 + NT - 1) / NT) : (NB * 81 + NT - 1) / NT, NLN = (NB * 6 * D2 + NT - 1) / NT;
   // U LINES AND JUMPS IN THE LOAD PHASE (round 6, whole-stage kernel of lo 5): the thread that loads a line of u contracts it along x for
   // all Q planes (U1) and the thread that loads a neighbour's trace value also loads the own face value (an L2 hit: this workgroup
   // reads the element anyway) and stores the jump -- the x-pencil phase of u, the trace step and their barrier are gone, u itself
   // never reaches LDS.  The same operations in the same order: bit-identical.  (The stall-dominated pencil and trace phases were 8 %
   // of a p = 3 workgroup's cycles, profiles/r06_phase_cycles.txt.)
   // Measured (one box, tools/kbench.py): p = 3 +0.5 ... 0.6 %, p = 5 +0.3 %, p = 4 +0.2 %, p = 6 +0.1 %, p = 2 -0.4 % (not there).
   // The lo 4 stage kernels (u also stored for the subcell pass, the barrier kept for the sub-mesh nodes): p = 3 +-0, p = 6 -0.15 %,
   // p = 4 +0.1 %, p = 5 +0.5 % -- not there.
   // (RMH_ULN is defined in front of K2Cfg: the stage kernel's layout drops the u slot with it)
   constexpr bool ULN = FUSED && !LO4 && (RMH_ULN);
   constexpr int NLU = ULN ? D * ((NB * D2 + (NT == 128 && NB == 1 ? 64 : NT) - 1) / (NT == 128 && NB == 1 ? 64 : NT)) : (NB * D3 + NT - 1) / NT;
   constexpr int NLS = (NB * 27 + NT - 1) / NT;
   const int nblk = (a.e_end - a.e_begin + NB - 1) / NB;
   int nbi[NLN], sti[NLS];
   double gx0[NLX], gv[NLX], gu[NLU];
   int itmax = 0, cg_known = 0;
   int blk = blockIdx.x;
#ifndef RMH_EARLY_EXIT
   // (every launch uses exactly one workgroup per batch, rmh_api.hip.  No early exit for blockIdx.x >= nblk: the test made the
   // kernel wait for e_begin / e_end -- a scalar round trip to the kernarg segment -- before it loaded any other argument; a
   // surplus workgroup would only redo the last batch, every store is guarded by e < e_end)
#else
   if (blk >= nblk) { return; }
   if (gridDim.x == (unsigned)nblk)
#endif
   {
      // workgroups are handed to the 8 XCDs round-robin (blockIdx.x % 8): give every XCD -- every L2 -- one contiguous
      // eighth of the element batches, so that the x- and y-neighbours whose traces and extrema an element reads were
      // touched by workgroups of the same XCD a few batches earlier (p = 3: HBM traffic per launch 3.71 -> 2.50 GB)
      const int xcd = blk & 7, j8 = blk >> 3, q8 = nblk >> 3, r8 = nblk & 7;
      // On a lattice (a.xcd_chunk = the batches of one z-layer) the eighths are cut into layers dealt round-robin: XCD k works
      // on layer 8 r + k in round r, so the +-z neighbours of its elements are in flight on XCD k +- 1 at the same time and the
      // second reader of a shared face-speed block / trace / extremum finds it in the Infinity Cache instead of HBM -- inside
      // a contiguous eighth it comes a whole layer (~55 MB of traffic at p = 3) later, far beyond the 4 MB L2.  Measured
      // (profiles/r05_xcd_layers.txt): p = 3 +2 %, p = 6 +3 %.  The batches behind the last whole round keep contiguous eighths.
      const int C_ = a.xcd_chunk;
      // (straight-line: with branches here the kernel-argument loads behind them waited for the branch -- two more scalar-memory
      // round trips in front of the first global load.  rounds = q8 / C_ and floor(2^32 / C_) come from the host: j8 / C_ by
      // multiplication, at most one below the quotient)
      const bool chunked = C_ > 0 && q8 >= C_;
      const int rounds = a.xcd_rounds, covered = rounds * 8 * C_;
      int jr = (int)__umulhi((unsigned)j8, a.xcd_inv);
      jr += ((jr + 1) * C_ <= j8) ? 1 : 0;
      const int jj = j8 - jr * C_, w = a.xcd_weave;
      const int woven = jr * 8 * C_ + xcd * C_ + (jj & ((1 << w) - 1)) * (C_ >> w) + (jj >> w); // (C_ is a multiple of 2^w)
      const int nt = nblk - covered, qt = nt >> 3, rt = nt & 7;
      const int tail = covered + xcd * qt + min(xcd, rt) + (j8 - rounds * C_);
      const int eighth = xcd * q8 + min(xcd, r8) + j8;
      blk = chunked ? (jr < rounds ? woven : tail) : eighth;
      blk = a.reverse ? nblk - 1 - blk : blk;
   }
   const int e0 = a.e_begin + blk * NB;
   const int tid = tid0;
   // (p = 6) split of the pencil-type phases over the two wavefronts of the workgroup, see split_outputs
   constexpr bool SPL = NT == 128 && NB == 1;
#if defined(__HIP_DEVICE_COMPILE__)
   const int wv = SPL ? __builtin_amdgcn_readfirstlane(tid >> 6) : 0;
#else
   const int wv = SPL ? tid >> 6 : 0;
#endif
   const int ptid = SPL ? (tid & 63) : tid; // task index of this thread in a split phase
   constexpr int PNT = SPL ? 64 : NT;
   // (p = 6) column split: 81 quadrature columns on two wavefronts left the second one a full pass over 17 columns.  The
   // first wavefront keeps columns 0..63; on the second, three lanes share a column -- a third of the qz range each, the
   // partial z-leg sums added across the three lanes (DPP row shifts) -- and it takes the face rows off the first one's hands.
   constexpr bool CSPL = SPL && C::CSPL;
   const int frt = CSPL ? (tid ^ 64) : tid; // face-row index of this thread (round 0)
   // opaque LDS row bases (opaque_lds_offset) where an element block is larger than a ds_read2_b64 reaches
#ifndef RMH_OPAQUE_BASE
#define RMH_OPAQUE_BASE (P >= 4)
#endif
   constexpr bool OPQ = RMH_OPAQUE_BASE;
#ifndef RMH_FACE_OPAQUE
#define RMH_FACE_OPAQUE (P != 5) // (p = 4, 6 +0.4 %, lo 4 at p = 6 +0.6 %; p = 5 -0.4 %: 8 B/lane more scratch)
#endif
   constexpr bool FOPQ = OPQ && RMH_FACE_OPAQUE; // (the same in the face rows)
#ifndef RMH_UNIFORM_SCALARS
#define RMH_UNIFORM_SCALARS 1
#endif
   constexpr bool UNI = NB == 1 && RMH_UNIFORM_SCALARS; // element scalars formed once per lane instead of once per dof round
   // transposed table rows where an output gathers a table column (TabLayoutQ::oBgT ...): from the order on at which the
   // table is read through views (below that it sits in scalar registers whole, and more of it would spill)
#ifndef RMH_TAB_TRANSPOSED
#define RMH_TAB_TRANSPOSED (P >= RMH_VIEW_MINP)
#endif
   constexpr bool TT = RMH_TAB_TRANSPOSED;
   // Outputs per table view in the pencil-type contractions (one output = one table row = one or two wide scalar loads).  With a
   // view per output the row of output k + 1 was loaded into the registers of row k, i.e. behind its FMAs: a scalar-memory round
   // trip per output with 4-9 FMAs in between (p = 5 back-transform: 18 exposed waits per task).  A group of rows loaded through
   // one view arrives together; sized to ~24 doubles (48 scalar registers) in flight.  GD: rows of D entries, GQ: of Q, G2D: two rows of D.
#ifndef RMH_VIEW_GROUP
#define RMH_VIEW_GROUP ((P == 5) ? 16 : 24) // (p = 5: 16 +1.2 %, 24 +0.7 %; p = 4, 6: 24 best; 36 loses at p = 5, 6)
#endif
   constexpr int GD = (P >= RMH_VIEW_MINP && RMH_VIEW_GROUP / D > 1) ? RMH_VIEW_GROUP / D : 1;
   constexpr int GQ = (P >= RMH_VIEW_MINP && RMH_VIEW_GROUP / Q > 1) ? RMH_VIEW_GROUP / Q : 1;
   constexpr int G2D = (P >= RMH_VIEW_MINP && RMH_VIEW_GROUP / (2 * D) > 1) ? RMH_VIEW_GROUP / (2 * D) : 1;
   // (the geometry pass of the column phase keeps its view per quadrature plane: a view per 2 or 3 planes -- 7 table entries each --
   // made the compiler keep whole groups of planes in flight: p = 6 22.9 k -> 10.2 k, p = 5 24.4 k -> 20.8 k MDOFs*stage/s)
   constexpr int G7 = 1;
#ifndef RMH_FACE_VIEW_GROUP
#define RMH_FACE_VIEW_GROUP 2 // (1 -> 2: p = 4, 5, 6 +0.7 ... +0.8 %)
#endif
   constexpr int GF = (P >= RMH_VIEW_MINP) ? RMH_FACE_VIEW_GROUP : 1;
   // pencil phases that handle several tensors per task read all their inputs before the first store (y-leg, x-leg)
#ifndef RMH_PRELOAD_LINES
#define RMH_PRELOAD_LINES (NB == 1 && NT == 128) // (p = 6 +0.3 %; p = 4, 5 -0.3 ... -0.8 %: one wavefront per workgroup keeps the tensor-by-tensor order)
#endif
   constexpr bool PRE = RMH_PRELOAD_LINES;
   typedef tabp_t<P> tabp;
   tabp gtb = (tabp)c_tab[P]; // constant memory: compile-time indices become scalar loads
   tabp gt = gtb;
   (void)gt;
   // One leg of a 1-D change of basis with the D x D table at offset oT, along direction dir, for every element of the batch: a
   // thread reads a line's D inputs once and forms its D outputs (split workgroups: half of them) with the table row as scalar
   // operands; in the dof role every output read D inputs and D table entries from LDS (p = 6: 42 instead of 7 LDS reads per
   // thread and direction; p = 3 +0.5 %, lo 4 +2.3 %).  Same sums in the same order.  The input of the first leg and the output
   // of the last (direction 2) are in the plain layout; the intermediates are stored with the z-planes JS apart: the lines of
   // direction 1 -- (ix, iz), inputs D apart -- otherwise start D^2 apart, i.e. in D banks (p = 3: 4-way conflicts, 18 % of the
   // stage's).  No barrier inside.
   auto basis_leg = [&](auto oT_, const int dir, const bool first, const int oin, const int oout) {
      constexpr int oT = decltype(oT_)::value, JS = C::JS;
      const int zin = first ? D2 : JS, zout = (dir == 2) ? D2 : JS; // plane strides of input and output
      const int sin_ = (dir == 0) ? 1 : (dir == 1 ? D : zin), sout = (dir == 0) ? 1 : (dir == 1 ? D : zout);
      for (int k0 = ptid; k0 < NB * D2; k0 += PNT)
      {
         const int eb = k0 / D2, k = k0 % D2;
         // line k: (iy, iz) = (k % D, k / D) along x; (ix, iz) along y; (ix, iy) along z
         const int bin = (dir == 0) ? D * (k % D) + zin * (k / D) : (dir == 1 ? (k % D) + zin * (k / D) : k);
         const int bout = (dir == 0) ? D * (k % D) + zout * (k / D) : (dir == 1 ? (k % D) + zout * (k / D) : k);
         const double *src = RMH_W(eb) + oin + bin;
         double in[D];
#pragma unroll
         for (int j = 0; j < D; j++) { in[j] = src[j * sin_]; }
         double *dst = RMH_W(eb) + oout + bout;
         split_outputs<SPL, D>(wv, [&](auto klo, auto khi) {
tabp gt = gtb;
#pragma unroll
            for (int kk = klo; kk < khi; kk++)
            {
               if (((kk) - (klo)) % GD == 0) { gt = RMH_TABK(); } // (one view per group of outputs: GD)
               double acc = 0.0;
#pragma unroll
               for (int j = 0; j < D; j++) { acc += gt[oT + kk * D + j] * in[j]; }
               dst[kk * sout] = acc;
            }
         });
      }
   };
#if defined(RMH_NOP_PROBE_A) && defined(__HIP_DEVICE_COMPILE__)
   // (diagnostic build only: the same probe in the load phase, see RMH_NOP_PROBE)
#pragma unroll
   for (int i_ = 0; i_ < RMH_NOP_PROBE_A; i_++) { asm volatile("v_nop"); }
#endif
   if (tid < 4 * NB) { s_acc[tid] = 0.0; } // reduction ring starts zeroed
   // ("any element still active" flags of the PCG loop: cleared here, in front of the first barrier -- the wavefronts reach
   // the prelude of the mass solve, where the first flag is raised, without a common barrier in between when their element
   // sums stay in the wavefront, RMH_WAVE_DOT)
   if (tid < 4) { s_flag[tid] = 0; }
   // all global loads are issued before the first LDS store so that they are in flight together
   // (neighbour indices first: the trace loads depend on them; with them the table slots of this thread's face rows)
   constexpr bool FMJ = RMH_FACE_MAJOR && NB > 1;
   constexpr int NFR = (NB * 6 * Q + NT - 1) / NT;
   int fri[NFR];
#pragma unroll
   for (int jp = 0; jp < NFR; jp++)
   {
      const int fr = min(frt + jp * NT, NB * 6 * Q - 1);
      const int feb = FMJ ? (fr % (NB * Q)) / Q : fr / (6 * Q), ff = FMJ ? fr / (NB * Q) : (fr % (6 * Q)) / Q;
      fri[jp] = a.face_rows[(size_t)min(e0 + feb, a.e_end - 1) * 6 + ff];
   }
   // trace table entries of this thread's face-layer entries (TabLayoutQ::oTr): independent loads, issued in front of the
   // neighbour indices they will be combined with
#ifndef RMH_TRACE_TABLE
#define RMH_TRACE_TABLE 1
#endif
   constexpr bool TRT = RMH_TRACE_TABLE;
   unsigned trpk[NLN];
#pragma unroll
   for (int j = 0; j < NLN; j++)
   {
      trpk[j] = 0u;
      if (TRT) { trpk[j] = ((const unsigned *)(a.tab + C::oTr))[min(tid + j * NT, NB * 6 * D2 - 1) % (6 * D2)]; }
   }
   load_batch<C, FUSED, ULN>(a, e0, tid, nbi, sti, gx0, gv, gu);
   // table copy for lane-dependent indexing: loaded behind the element data, stored with it (a copy loop at the top of
   // the kernel put a full memory round trip in front of the first element load)
   constexpr int NLT = (C::N2S + NT - 1) / NT;
   double gtab[NLT];
#pragma unroll
   for (int j = 0; j < NLT; j++) { gtab[j] = a.tab[min(tid + j * NT, C::N2S - 1)]; }
   // RD solver: sub-mesh start positions and node velocities, in flight with everything else (a load-and-store loop
   // in front of the first barrier exposed five memory round trips: 18 % of the lo 4 workgroup's cycles)
   constexpr int NLSUB = LO4 ? (NB * 3 * D3 + NT - 1) / NT : 1;
   double gsx[NLSUB], gsv[NLSUB];
   auto load_submesh = [&]() {
      if (LO4 && a.rd_subcell)
      {
#pragma unroll
         for (int j = 0; j < NLSUB; j++)
         {
            const int k = min(tid + j * NT, NB * 3 * D3 - 1);
            const size_t g = (size_t)min(e0 + k / (3 * D3), a.e_end - 1) * 3 * D3 + k % (3 * D3);
            gsx[j] = a.subx0[g];
            gsv[j] = a.move ? a.subvel[g] : 0.0;
         }
      }
   };
   // SUBL (p = 3): the sub-mesh nodes are the youngest loads and go to LDS behind the u pencils (one more barrier) -- the first
   // barrier then waits for the nodes, u and the tables only: lo 4 +1.2 % at p = 3 (p = 4 +-0, p = 6 -0.8 %: not there)
   constexpr bool SUBL = LO4 && P == 3;
   if (!SUBL) { load_submesh(); }
   // ... and the sub-mesh velocities at the subcell midpoints of this thread's subcells (read inside the subcell pass they
   // were a memory round trip in the open)
   constexpr int NSR = LO4 ? (NB * C::NS + NT - 1) / NT : 1;
   double gvm[NSR][3];
   if (LO4)
   {
#pragma unroll
      for (int j = 0; j < NSR; j++)
      {
         const int k = min(tid + j * NT, NB * C::NS - 1);
         const double *vmid = a.subvmid + (size_t)min(e0 + k / C::NS, a.e_end - 1) * 3 * C::NS + k % C::NS;
#pragma unroll
         for (int comp = 0; comp < 3; comp++) { gvm[j][comp] = a.rd_subcell ? vmid[comp * C::NS] : 0.0; }
      }
   }
   double gn[NLN], go[ULN ? NLN : 1];
   // LDS offsets of the own face dof and of the jump slot of this thread's trace entries: the trace step of phase B took them apart
   // again (entry -> element, face, face dof, strides: ~30 integer instructions per entry; kept: p = 4, 5, 6 +0.9 ... +1.2 %, p = 3 +0.7 %)
   int tr_own[NLN], tr_dst[NLN];
#pragma unroll
   for (int j = 0; j < NLN; j++)
   {
      // (straight-line like load_batch: boundary faces and lanes past the list load a valid entry and drop it)
      const int k = min(tid + j * NT, NB * 6 * D2 - 1);
      {
         const int r6 = k % (6 * D2);
         int own_off, nbr_off, r;
         if (TRT)
         {
            // (own face dof | opposite-layer dof of the neighbour | index in the layer: from the trace table)
            own_off = (int)(trpk[j] & 1023u);
            nbr_off = (int)((trpk[j] >> 10) & 1023u);
            r = (int)(trpk[j] >> 20);
         }
         else
         {
            const int f = r6 / D2;
            r = r6 % D2;
            const int i1 = r % D, i2 = r / D;
            const int c = f >> 1, side = f & 1;
            const int strc = axis_stride<D>(c), str1 = axis_stride<D>(axis_next(c)), str2 = axis_stride<D>(axis_next2(c));
            own_off = (side ? P * strc : 0) + i1 * str1 + i2 * str2;
            nbr_off = (side ? 0 : P) * strc + i1 * str1 + i2 * str2;
         }
         tr_own[j] = (k / (6 * D2)) * C::EL + oU + own_off;
         tr_dst[j] = (k / (6 * D2)) * C::EL + oNb + r6;
         const int nb = max(nbi[j], 0);
         const double *un = (nb < a.ne_owned) ? a.u + (size_t)nb * D3 : a.u_ghost + (size_t)(nb - a.ne_owned) * a.gh_ustride;
         // the neighbour's opposite face layer (compact ghost records hold exactly that layer, ordered like this face:
         // rmh_exchange_setup)
         const int off = (a.gh_compact && nb >= a.ne_owned) ? r : nbr_off;
         const double v = un[off];
         gn[j] = nbi[j] >= 0 ? v : 0.0; // boundary: u_nbr = 0 (no inflow data enters the HO path)
         if (ULN) { go[ULN ? j : 0] = a.u[(size_t)min(e0 + k / (6 * D2), a.e_end - 1) * D3 + own_off]; }
      }
   }
   // face speed coefficients of this thread's face rows (youngest loads: first used after the second barrier)
   // face-major order of the face rows of a multi-element workgroup: row index = (face, element, q1).  The lanes of an LDS
   // access group then work on ONE face of several elements -- the same node and trace offsets, element blocks apart (EL == 2
   // mod 32: different banks) -- instead of on several faces of one element, whose trace blocks (D^2 doubles apart) and face
   // nodes share banks (tools/pmc_variants.sh: the face rows were 30 % of the p = 3 stage's bank conflicts)
   constexpr bool HX = (RMH_HIER & 1) != 0, HY = (RMH_HIER & 2) != 0, HZ = (RMH_HIER & 4) != 0; // hierarchical directions of the mesh nodes
   double fgc[NFR][3 * Q];
   {
#pragma unroll
      for (int jp = 0; jp < NFR; jp++)
      {
         const int fr = min(frt + jp * NT, NB * 6 * Q - 1);
         // (the block of this row's face: the element's own, or the face neighbour's with the sign flipped -- FaceGeo)
         const double *fg = a.fgeo + (size_t)(fri[jp] & ~RMH_FACE_FLIP) * FaceGeo<P>::SLOT + fr % Q;
         // (a static mesh -- transport -- has c1 = c2 = 0: their loads are pointed at the c0 row, which is in the cache anyway, and
         // the face rows multiply them by zero; straight-line loads either way: a.move is uniform over the launch)
         // (+3.8 % for transport at p = 3, remap +-0; not at p >= 4, where the extra scalar arithmetic of the 21-27 loads costs remap 0.4-0.7 %)
         const int mv = (a.move || P >= 4) ? 1 : 0;
         // (three base pointers, compile-time offsets: with the index (k - (1 - mv) (k % 3)) Q formed per load, 12 of the 18 loads of a
         // row carried a 64-bit address add of their own)
         const double *fgk[3] = {fg, fg - (1 - mv) * Q, fg - 2 * (1 - mv) * Q};
#pragma unroll
         for (int k = 0; k < 3 * Q; k++) { fgc[jp][k] = fgk[k % 3][k * Q]; }
      }
   }
   // diagnostic: the largest iteration count so far, read here -- behind the element loads, a uniform load whose
   // latency is covered by theirs -- so that a workgroup that does not raise it issues no atomic and waits for nothing
   // (a stale value only costs a redundant atomicMax)
   cg_known = HAS_HO ? *a.cg_iters : 0;
   if (SUBL) { load_submesh(); }
   if constexpr (C::XPK != 0)
   {
      tabp gt = RMH_TAB();
#pragma unroll
      for (int j = 0; j < NLX / 3; j++)
      {
         const int k = tid + j * NT;
         if (k < NB * 27)
         {
            const int eb = k / 27, l = k % 27;
            const double x0n = a.move ? gx0[3 * j] + a.t * gv[3 * j] : gx0[3 * j];
            const double x1n = a.move ? gx0[3 * j + 1] + a.t * gv[3 * j + 1] : gx0[3 * j + 1];
            const double x2n = a.move ? gx0[3 * j + 2] + a.t * gv[3 * j + 2] : gx0[3 * j + 2];
            const double v0n = gv[3 * j], v1n = gv[3 * j + 1], v2n = gv[3 * j + 2];
            // ([kind][qx][line]: the 27 line threads of an element store side by side, the columns read their plane 27 doubles apart --
            // banks 54 dwords apart; [line][kind][qx] had 3-way conflicts on every store: SQ_LDS_BANK_CONFLICT 8.0e7 -> 1.29e8 at p = 3)
            double *XP = RMH_W(eb) + oXV + l;
#pragma unroll
            for (int qx = 0; qx < Q; qx++)
            {
               // (the column's own expressions, see column_pass)
               const double L1 = gt[oL + qx * 3 + 1], L2 = gt[oL + qx * 3 + 2];
               XP[qx * 27] = x0n + L1 * x1n + L2 * x2n;
               XP[(Q + qx) * 27] = v0n + L1 * v1n + L2 * v2n;
               if (C::XPK == 3) { XP[(2 * Q + qx) * 27] = gt[odL + qx * 3 + 1] * x1n + gt[odL + qx * 3 + 2] * x2n; }
            }
            if (C::XPK == 2)
            {
               RMH_W(eb)[C::oXR + 2 * l] = x1n;
               RMH_W(eb)[C::oXR + 2 * l + 1] = x2n;
            }
         }
      }
   }
   else
   {
#pragma unroll
      for (int j = 0; j < NLX; j++)
      {
         const int k = tid + j * NT;
         if (k < NB * 81)
         {
            const int eb = k / 81, i = k % 81;
            RMH_W(eb)[oXV + 81 + i] = gv[j];
            RMH_W(eb)[oXV + i] = a.move ? gx0[j] + a.t * gv[j] : gx0[j];
         }
      }
   }
   if constexpr (ULN)
   {
      // U1[eb][(kind*Q + qx)*S2 + i2], kind 0: B.u, 1: G.u, from the line in registers (the pencil phase's own expressions)
#pragma unroll
      for (int j = 0; j < NLU / D; j++)
      {
         const int k = ptid + j * PNT;
         if (k < NB * D2)
         {
            double *dst = RMH_W(k / D2) + oU1 + k % D2;
            split_outputs<SPL, Q>(wv, [&](auto qlo, auto qhi) {
               constexpr bool PV3 = (P == 3) && P < RMH_VIEW_MINP;
               std::conditional_t<PV3, tabp_const, tabp> gt = (std::conditional_t<PV3, tabp_const, tabp>)gtb;
               if constexpr (PV3) { gt = tab_view_c<P>(); }
#pragma unroll
               for (int q = qlo; q < qhi; q++)
               {
                  if constexpr (!PV3) { if (((q) - (qlo)) % G2D == 0) { gt = RMH_TABK(); } }
                  double ub = 0.0, ug = 0.0;
#pragma unroll
                  for (int ix = 0; ix < D; ix++)
                  {
                     ub += gt[oB + q * D + ix] * gu[j * D + ix];
                     ug += gt[oG + q * D + ix] * gu[j * D + ix];
                  }
                  dst[(0 * Q + q) * S2] = ub;
                  dst[(1 * Q + q) * S2] = ug;
               }
            });
         }
      }
      // the jumps u_nbr - u_own at the face dofs (see the trace step of phase B)
#pragma unroll
      for (int j = 0; j < NLN; j++)
      {
         if (tid + j * NT < NB * 6 * D2) { lds[tr_dst[j]] = gn[j] - go[ULN ? j : 0]; }
      }
   }
   else
   {
#pragma unroll
      for (int j = 0; j < NLU; j++)
      {
         const int k = tid + j * NT;
         if (k < NB * D3) { RMH_W(k / D3)[oU + k % D3] = gu[j]; }
      }
   }
#pragma unroll
   for (int j = 0; j < NLT; j++) { if (tid + j * NT < C::N2S) { stab[tid + j * NT] = gtab[j]; } }
   if (FUSED)
   {
      // (in a register the index would be spilled through the column phase, and a pending scratch reload makes
      // the compiler drain ALL outstanding loads before the extrema loads of the PCG prelude can issue)
#pragma unroll
      for (int j = 0; j < NLS; j++)
      {
         const int k = tid + j * NT;
         if (k < NB * 27) { s_sti[k] = sti[j]; }
      }
   }
   auto store_submesh = [&]() {
      if (LO4 && a.rd_subcell)
      {
         // sub-mesh nodes x_sub(t) = x0_sub + t v_sub (remhos.cpp:1262-1274); x0_sub was set up once
#pragma unroll
         for (int j = 0; j < NLSUB; j++)
         {
            const int k = tid + j * NT;
            if (k < NB * 3 * D3) { RMH_W(k / (3 * D3))[C::oXs + k % (3 * D3)] = a.move ? fma(a.t, gsv[j], gsx[j]) : gsx[j]; }
         }
      }
   };
   if (!SUBL) { store_submesh(); }
   __syncthreads();

   RMH_STAMP(0);
   // ---- phase B: x-contractions of the geometry and of u; face rows -------------------------------
   // element extrema (remhos_tools.cpp:497-523)
   double my_min = INFINITY, my_max = -INFINITY;
   if (FUSED || LO4) {}
   else if (C::WAVE_ALIGNED)
   {
      // D3 = 64: the values a wavefront loaded in round j all belong to element (j*NT + tid)/D3
#pragma unroll
      for (int j = 0; j < NLU; j++)
      {
         const int k = tid + j * NT;
         const double lo = wave_minmax<true>(k < NB * D3 ? gu[j] : INFINITY);
         const double hi = wave_minmax<false>(k < NB * D3 ? gu[j] : -INFINITY);
         if ((tid & 63) == 63 && k < NB * D3 && e0 + k / D3 < a.e_end)
         {
            a.xe_min[e0 + k / D3] = lo;
            a.xe_max[e0 + k / D3] = hi;
         }
      }
   }
   else if (!FUSED && !LO4 && NB == 1)
   {
      // one element per workgroup (p = 4, 5, 6): from the loaded values in registers -- thread, wavefront (DPP), and the
      // wavefront results in the spare doubles behind the flags; thread 0 combines them at the end of the kernel (a
      // serial loop of one thread over the D^3 values in LDS was 22 k cycles at p = 6)
#pragma unroll
      for (int j = 0; j < NLU; j++)
      {
         const bool in = tid + j * NT < D3;
         my_min = fmin(my_min, in ? gu[j] : INFINITY);
         my_max = fmax(my_max, in ? gu[j] : -INFINITY);
      }
      constexpr int NW = NT / 64;
      double *part = s_acc + 4 * NB + 2; // (8 doubles are reserved for the flags, the first two hold them)
      static_assert(NB != 1 || NT / 64 <= 3, "spare doubles behind the flags");
      if (RMH_PACKED_SUMS)
      {
         const double mm = wave_min_negmax(my_min, my_max);
         if ((tid & 63) == 31) { part[2 * (tid >> 6)] = mm; }
         if ((tid & 63) == 63) { part[2 * (tid >> 6) + 1] = -mm; }
      }
      else
      {
         my_min = wave_minmax<true>(my_min);
         my_max = wave_minmax<false>(my_max);
         if ((tid & 63) == 63)
         {
            part[2 * (tid >> 6)] = my_min;
            part[2 * (tid >> 6) + 1] = my_max;
         }
      }
   }
   else if (!FUSED && !LO4 && tid < NB)
   {
      const double *uu = RMH_W(tid) + oU;
      for (int i = 0; i < D3; i++)
      {
         my_min = fmin(my_min, uu[i]);
         my_max = fmax(my_max, uu[i]);
      }
   }
   // U1[eb][(kind*Q + qx)*S2 + i2], kind 0: B.u, 1: G.u; pencil tasks (eb, i2)
   static_assert(!SPL || NB * D2 <= 64, "split phases: one task per lane");
   for (int k = ptid; k < (ULN ? 0 : NB * D2); k += PNT)
   {
      const int eb = k / D2, i2 = k % D2;
      const double *src = RMH_W(eb) + oU + D * i2;
      double uu[D];
#pragma unroll
      for (int ix = 0; ix < D; ix++) { uu[ix] = src[ix]; }
      double *dst = RMH_W(eb) + oU1 + i2;
      split_outputs<SPL, Q>(wv, [&](auto qlo, auto qhi) {
#ifndef RMH_PENCIL_VIEW3
#define RMH_PENCIL_VIEW3 (P == 3) // (B and G, 2 x 24 doubles: 13 v_writelane + 12 v_readlane per wavefront otherwise; see the y-leg)
#endif
         constexpr bool PV3 = RMH_PENCIL_VIEW3 && P < RMH_VIEW_MINP;
         std::conditional_t<PV3, tabp_const, tabp> gt = (std::conditional_t<PV3, tabp_const, tabp>)gtb;
         if constexpr (PV3) { gt = tab_view_c<P>(); }
#pragma unroll
         for (int q = qlo; q < qhi; q++)
         {
            if constexpr (!PV3) { if (((q) - (qlo)) % G2D == 0) { gt = RMH_TABK(); } } // (one view per group of outputs: GD)
            double ub = 0.0, ug = 0.0;
#pragma unroll
            for (int ix = 0; ix < D; ix++)
            {
               ub += gt[oB + q * D + ix] * uu[ix];
               ug += gt[oG + q * D + ix] * uu[ix];
            }
            dst[(0 * Q + q) * S2] = ub;
            dst[(1 * Q + q) * S2] = ug;
         }
      });
   }
   RMH_STAMP(1);
   if (SUBL)
   {
      store_submesh();
      __syncthreads();
   }
   if (LO4)
   {
      // (before the face rows: their Bernstein-tested rows reuse the LDS of the sub-mesh nodes)
      constexpr int NS = C::NS;
      const double eps = 1.E-15;
      // subcell fluctuations with the 1-point rule on the trilinear subcells and subcell extrema
      // (SetupSubCellPA3D / ApplySubCellWeights remhos_lo.cpp:1137-1192, 1313-1618; :1733-1757)
#pragma unroll
      for (int js = 0; js < NSR; js++)
      {
         const int k = tid + js * NT;
         if (k >= NB * NS) { break; }
         const int eb = k / NS, m = k % NS;
         const int mx = m % P, my = (m / P) % P, mz = m / (P * P);
         const int base = mx + D * my + D2 * mz;
         const double *su_ = RMH_W(eb) + oU, *xs = RMH_W(eb) + C::oXs;
         double J[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, vm[3] = {0, 0, 0};
         double umax = -INFINITY, umin = INFINITY, usum = 0.0;
#pragma unroll
         for (int j = 0; j < 8; j++)
         {
            const int i = base + (j & 1) + D * ((j >> 1) & 1) + D2 * (j >> 2);
            const double uj = su_[i];
            umax = fmax(umax, uj);
            umin = fmin(umin, uj);
            usum += uj;
#pragma unroll
            for (int comp = 0; comp < 3; comp++)
            {
               // plain residual distribution (lo 3, remhos_lo.cpp:965-1034) = the subcell scheme with
               // zero subcell fluctuations: the sub-mesh is not loaded and contributes nothing
               const double x = a.rd_subcell ? xs[comp * D3 + i] : 0.0;
               J[comp][0] += ((j & 1) ? 0.25 : -0.25) * x;
               J[comp][1] += ((j & 2) ? 0.25 : -0.25) * x;
               J[comp][2] += ((j & 4) ? 0.25 : -0.25) * x;
            }
         }
#pragma unroll
         for (int comp = 0; comp < 3; comp++) { vm[comp] = gvm[js][comp]; }
         const double A11 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
         const double A12 = J[2][1] * J[0][2] - J[0][1] * J[2][2];
         const double A13 = J[0][1] * J[1][2] - J[1][1] * J[0][2];
         const double A21 = J[2][0] * J[1][2] - J[1][0] * J[2][2];
         const double A22 = J[0][0] * J[2][2] - J[0][2] * J[2][0];
         const double A23 = J[1][0] * J[0][2] - J[0][0] * J[1][2];
         const double A31 = J[1][0] * J[2][1] - J[2][0] * J[1][1];
         const double A32 = J[2][0] * J[0][1] - J[0][0] * J[2][1];
         const double A33 = J[0][0] * J[1][1] - J[0][1] * J[1][0];
         const double q0 = a.alpha * (A11 * vm[0] + A12 * vm[1] + A13 * vm[2]);
         const double q1v = a.alpha * (A21 * vm[0] + A22 * vm[1] + A23 * vm[2]);
         const double q2v = a.alpha * (A31 * vm[0] + A32 * vm[1] + A33 * vm[2]);
         double fluct = 0.0;
#pragma unroll
         for (int j = 0; j < 8; j++)
         {
            const int i = base + (j & 1) + D * ((j >> 1) & 1) + D2 * (j >> 2);
            const double w = ((j & 1) ? 0.25 : -0.25) * q0 + ((j & 2) ? 0.25 : -0.25) * q1v + ((j & 4) ? 0.25 : -0.25) * q2v;
            fluct += w * su_[i];
         }
         double *fl = RMH_W(eb) + C::oSub;
         // [fluct | bound | ratio]: fluct^+ = max(0, fluct), fluct^- = min(0, fluct) are re-formed by the readers;
         // eqs. (58)-(59): the ratio fluct^+- / sumWeightsSubcell^+- is formed once per subcell, and only the one of
         // the two with a non-zero numerator is kept (the other contributes exactly zero) together with the bound it
         // multiplies -- the subcell maximum for fluct > 0, else the minimum.  Both ratios are >= 0: the second is
         // stored negated, its sign tells the dofs which of their two sums it belongs to.
         fl[0 * NS + m] = fluct;
         fl[1 * NS + m] = (fluct > 0.) ? umax : umin;
         // (one division: the numerators are fluct itself in both cases)
         const double ratio = fdiv(fluct, (fluct > 0.) ? 8 * umax - usum + eps : 8 * umin - usum - eps);
         fl[2 * NS + m] = (fluct > 0.) ? ratio : -ratio;
      }
   }
   RMH_STAMP(24);
   // The neighbour traces are the only loads that depend on another load (the neighbour index): they are not waited
   // for at the first barrier but here, behind the x-pencils of u, which need none of them.
   // What is stored is the JUMP u_nbr - u_own at the face dof: the Q face rows of a face each formed the same D^2
   // differences from two LDS reads apiece (p = 6: 98 reads and 49 subtractions per row, on the wavefront that is the
   // longer pole of the workgroup); now one read and one subtraction per trace value here, D^2 reads per row there.
   if constexpr (!ULN)
   {
#pragma unroll
      for (int j = 0; j < NLN; j++)
      {
         const int k = tid + j * NT;
         if (k < NB * 6 * D2)
         {
            lds[tr_dst[j]] = gn[j] - lds[tr_own[j]];
         }
      }
      __syncthreads();
   }
   RMH_STAMP(25);
   const double t_move = (a.move || P >= 4) ? a.t : 0.0; // (static mesh: the face speed is its value at t = 0; p >= 4 reads the zero coefficients)
   // face rows: thread (eb, f, q1) integrates the quadrature row {(q1, q2)} of face f:
   //   val(q) = w_q max(0, upw * v.n_out) (u_nbr - u_own)(q)      
   // and tests it along q2 with the GL nodal basis -> sFq[eb][(f*Q + q1)*D + k2]
#pragma unroll
   for (int jp = 0; jp < NFR; jp++)
   {
      const int fr = frt + jp * NT;
      if (fr >= NB * 6 * Q) { break; }
      const int eb = FMJ ? (fr % (NB * Q)) / Q : fr / (6 * Q);
      const int f = FMJ ? fr / (NB * Q) : (fr % (6 * Q)) / Q, q1 = fr % Q;
      // traces: contraction of the jumps u_nbr - u_own along i1
      // (OPQ: the lane-dependent bases of the trace block, the table row and the output rows as opaque offsets, opaque_lds_offset)
      const double *un = FOPQ ? lds + opaque_lds_offset((int)(RMH_W(eb) - lds) + oNb + f * D2) : RMH_W(eb) + oNb + f * D2;
      const double *tb1 = FOPQ ? lds + opaque_lds_offset((int)(stab - lds) + oB + q1 * D) : stab + oB + q1 * D;
      double jr[D];
#pragma unroll
      for (int i2 = 0; i2 < D; i2++)
      {
         double acc = 0.0;
#pragma unroll
         for (int i1 = 0; i1 < D; i1++) { acc += tb1[i1] * un[i1 + D * i2]; }
         jr[i2] = acc;
      }
      double tq[D], tq2[D];
#pragma unroll
      for (int k2 = 0; k2 < D; k2++) { tq[k2] = 0.0; tq2[k2] = 0.0; }
      const double upw_row = (fri[jp] < 0) ? -a.upw : a.upw; // (the neighbour's block of this face: opposite normal, FaceGeo)
      tabp gt = gtb;
#pragma unroll
      for (int q2 = 0; q2 < Q; q2++)
      {
         if (q2 % GF == 0) { gt = RMH_TABK(); } // (rows of B and Bg of GF quadrature points per view)
         // w_q1 w_q2 max(0, upw * v.n_out) at time t: the face speed is a quadratic in t (face_geom_kernel)
         const double sq = fmax(0.0, upw_row * (fgc[jp][3 * q2] + t_move * (fgc[jp][3 * q2 + 1] + a.t * fgc[jp][3 * q2 + 2])));
         double jump = 0.0;
#pragma unroll
         for (int i2 = 0; i2 < D; i2++) { jump += gt[oB + q2 * D + i2] * jr[i2]; }
         if (HAS_HO)
         {
            const double val = sq * jump;
#pragma unroll
            for (int k2 = 0; k2 < D; k2++) { tq[k2] += gt[oBg + q2 * D + k2] * val; }
         }
         if (LO4)
         {
#pragma unroll
            for (int k2 = 0; k2 < D; k2++) { tq2[k2] += gt[oB + q2 * D + k2] * sq; }
         }
      }
      double *frow_out = FOPQ ? lds + opaque_lds_offset((int)(RMH_W(eb) - lds) + (f * Q + q1) * D) : RMH_W(eb) + (f * Q + q1) * D;
      if (HAS_HO)
      {
#pragma unroll
         for (int k2 = 0; k2 < D; k2++) { frow_out[oF + k2] = tq[k2]; }
      }
      if (LO4)
      {
#pragma unroll
         for (int k2 = 0; k2 < D; k2++) { frow_out[C::oF2 + k2] = tq2[k2]; }
      }
   }
   RMH_STAMP(26);
   // (no barrier: the column phase reads the nodes and U1, both complete since the barrier above; the face rows' output is
   // read after the next one.  Split columns, p = 6: the face rows run on the second wavefront BESIDE the first one's column pass)

   // lumped upwind face fluxes of the RD solver (ApplyFaceTerms3D, remhos_lo.cpp:795-871): (B^T D B 1)_i (u_nbr,i - u_i) with
   // the face rows already tested along q2.  Pass 1, tasks (element, face, line i2): the Q tested rows of the line are read
   // once and contracted along q1 for its D face dofs (table rows as scalar operands); the coefficient times the jump
   // replaces the jump.  Pass 2, dof role: a dof adds the products of the (up to three) faces it lies on, in the order of
   // the axes.  (Gathered per dof in one pass, every face dof read its Q rows and Q table entries from LDS: p = 6 2.9 k LDS
   // reads per element instead of 1.0 k.)
   auto lumped_face_products = [&]() {
      for (int k = ptid; k < NB * 6 * D; k += PNT)
      {
         const int eb = k / (6 * D), rem = k % (6 * D);
         const int f = rem / D, i2 = rem % D;
         const double *F = RMH_W(eb) + C::oF2 + f * Q * D + i2;
         double in[Q];
#pragma unroll
         for (int q1 = 0; q1 < Q; q1++) { in[q1] = F[q1 * D]; }
         double *tr = RMH_W(eb) + oNb + f * D2 + D * i2;
         split_outputs<SPL, D>(wv, [&](auto lo_, auto hi_) {
tabp gt = gtb;
#pragma unroll
            for (int i1 = lo_; i1 < hi_; i1++)
            {
               if (((i1) - (lo_)) % GQ == 0) { gt = RMH_TABK(); } // (one view per group of outputs: GD)
               double coef = 0.0;
#pragma unroll
               for (int q1 = 0; q1 < Q; q1++) { coef += (TT ? gt[C::oBT + i1 * Q + q1] : gt[oB + q1 * D + i1]) * in[q1]; }
               tr[i1] = coef * tr[i1];
            }
         });
      }
   };
   auto lumped_face_gather = [&]() {
      for (int t = tid; t < NB * D3; t += NT)
      {
         const int eb = t / D3, i = t % D3;
         const int idx[3] = {i % D, (i / D) % D, i / D2};
         double acc = 0.0;
#pragma unroll
         for (int c = 0; c < 3; c++)
         {
            const int ic = idx[c];
            if (ic == 0 || ic == P)
            {
               const int c1 = (c + 1) % 3, c2 = (c + 2) % 3;
               const int f = 2 * c + (ic == P ? 1 : 0);
               acc += RMH_W(eb)[oNb + f * D2 + idx[c1] + D * idx[c2]];
            }
         }
         RMH_W(eb)[C::oDuf + i] = acc;
      }
   };
   RMH_STAMP(2);
   // ---- phase C: column threads: geometry, grad u, z-leg of the test contractions -------------------
   // column role: thread -> quadrature column cc = qx + Q qy; in the split wavefront (CSPL) three lanes share a column, lane
   // 3 c + zt takes the third zt of its qz range
   const int sl = tid - 64;
   const bool zsplit = CSPL && wv == 1;
   // (split wavefront: five triples per 16-lane row, the sixteenth lane idle -- the three partial sums of a column then
   // meet by DPP row shifts; with the triples packed across rows they needed ds_bpermute)
   static_assert(!CSPL || 20 >= Q2 - 64, "split columns: four rows of five triples");
   const int srow = (sl >> 4) & 3, sin = sl & 15;
   const int scc = 64 + srow * 5 + sin / 3;
   const bool col = CSPL ? (tid < 64 || (sin < 15 && scc < Q2)) : tid < NB * Q2;
   const int ceb = (col && !CSPL) ? tid / Q2 : 0;
   const int cc = CSPL ? (tid < 64 ? tid : min(scc, Q2 - 1)) : tid % Q2;
   const int zt = zsplit ? (sin % 3) % 3 : 0;
   const int qx = cc % Q, qy = cc / Q;
   double wd[Q];
   double Bgy[D]; // GL basis row of this thread's qy (mass apply)
   double r0[D], r1[D], r2[D];
#pragma unroll
   for (int iz = 0; iz < D; iz++) { r0[iz] = 0; r1[iz] = 0; r2[iz] = 0; Bgy[iz] = 0; }
#pragma unroll
   for (int qz = 0; qz < Q; qz++) { wd[qz] = 0; }
   // ZS: the split form -- NQ = Q / 3 quadrature points per lane, table rows of qz = zt NQ + j from the LDS copy (the index
   // depends on the lane); otherwise the whole column with compile-time rows (scalar operands)
   auto column_pass = [&](auto zs_) {
      constexpr bool ZS = decltype(zs_)::value;
      constexpr int NQ = ZS ? Q / 3 : Q;
      const int zo1 = ZS ? zt * NQ : 0, zo3 = 3 * zo1, zoD = D * zo1;
      // (split form: one opaque base per kind of table row -- the lane part zt is in it, the offsets that remain are the table's)
      const int sto = (int)(stab - lds);
      const double *z3 = (ZS && OPQ) ? lds + opaque_lds_offset(sto + zo3) : stab + zo3;
      const double *zD = (ZS && OPQ) ? lds + opaque_lds_offset(sto + zoD) : stab + zoD;
      const double *z1 = (ZS && OPQ) ? lds + opaque_lds_offset(sto + zo1) : stab + zo1;
      auto T3 = [&](tabp gt, int o, int qz, int i) { return ZS ? z3[o + qz * 3 + i] : gt[o + qz * 3 + i]; };
      auto TD = [&](tabp gt, int o, int qz, int i) { return ZS ? zD[o + qz * D + i] : gt[o + qz * D + i]; };
      auto T1 = [&](tabp gt, int o, int qz) { return ZS ? z1[o + qz] : gt[o + qz]; };
      // (RMH_OPAQUE_BASE: row bases as opaque offsets, see opaque_lds_offset)
      const double *tLy = OPQ ? lds + opaque_lds_offset((int)(stab - lds) + oL + qy * 3) : stab + oL + qy * 3;
      const double *tLx = OPQ ? lds + opaque_lds_offset((int)(stab - lds) + oL + qx * 3) : stab + oL + qx * 3;
      double Ly[3], dLy[3];
#pragma unroll
      for (int k = 0; k < 3; k++) { Ly[k] = tLy[k]; dLy[k] = tLy[odL - oL + k]; }
      double Lx[3], dLx[3];
#pragma unroll
      for (int k = 0; k < 3; k++) { Lx[k] = tLx[k]; dLx[k] = tLx[odL - oL + k]; }
      const double wxy = stab[oW + qx] * stab[oW + qy];
      double Dq[3][NQ], wl[NQ];
      {
         // pass 1: geometry.  x- and y-contractions of the 27 nodes of X(t) and V for this column
         // (broadcast LDS reads: all columns of an element read the same node)
         const double *XN = RMH_W(ceb) + oXV;
         const double *XPq = OPQ ? lds + opaque_lds_offset((int)(RMH_W(ceb) - lds) + oXV + qx * 27) : XN + qx * 27; // (XPK: the values of plane qx)
         double A[3][4][3];
#pragma unroll
         for (int comp = 0; comp < 3; comp++)
         {
#pragma unroll
            for (int az = 0; az < 3; az++)
            {
               double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
               for (int ay = 0; ay < 3; ay++)
               {
                  double xl, xd, vl;
                  if constexpr (C::XPK != 0)
                  {
                     // (contracted along x by the thread that loaded the line, phase A; K2Cfg::XPK)
                     constexpr int XK = C::XPK;
                     const int l = comp * 9 + ay + 3 * az;
                     xl = XPq[l];
                     vl = XPq[Q * 27 + l];
                     if (XK == 3) { xd = XPq[2 * Q * 27 + l]; }
                     else { xd = dLx[1] * XN[C::oXR + 2 * l] + dLx[2] * XN[C::oXR + 2 * l + 1]; }
                  }
                  else
                  {
                  const double *xr = XN + comp * 27 + 3 * (ay + 3 * az);
                  const double x0n = xr[0], x1n = xr[1], x2n = xr[2];
                  const double v0n = xr[81], v1n = xr[82], v2n = xr[83];
                  // (hierarchical nodes, RMH_HIER: basis (1, L1, L2), derivative (0, dL1, dL2), along x and along y)
                  xl = HX ? x0n + Lx[1] * x1n + Lx[2] * x2n : Lx[0] * x0n + Lx[1] * x1n + Lx[2] * x2n;
                  xd = HX ? dLx[1] * x1n + dLx[2] * x2n : dLx[0] * x0n + dLx[1] * x1n + dLx[2] * x2n;
                  vl = HX ? v0n + Lx[1] * v1n + Lx[2] * v2n : Lx[0] * v0n + Lx[1] * v1n + Lx[2] * v2n;
                  }
                  if (HY && ay == 0) { a0 = xd; a2 = xl; a3 = vl; }
                  else
                  {
                     a0 += Ly[ay] * xd;
                     a1 += dLy[ay] * xl;
                     a2 += Ly[ay] * xl;
                     a3 += Ly[ay] * vl;
                  }
               }
               A[comp][0][az] = a0; A[comp][1][az] = a1; A[comp][2][az] = a2; A[comp][3][az] = a3;
            }
         }
tabp gt = gtb;
#pragma unroll
         for (int qz = 0; qz < NQ; qz++)
         {
            if (ZS) { sched_fence(); }
            if (qz % G7 == 0) { gt = RMH_TABK(); } // (geometry pass: 7 table entries per plane -- a view per group of planes)
            double J[3][3], v[3];
#pragma unroll
            for (int comp = 0; comp < 3; comp++)
            {
               double j0 = 0, j1 = 0, j2 = 0, vv = 0;
               if (HZ) { j0 = A[comp][0][0]; j1 = A[comp][1][0]; vv = A[comp][3][0]; }
#pragma unroll
               for (int az = HZ ? 1 : 0; az < 3; az++)
               {
                  const double Lz = T3(gt, oL, qz, az), dLz = T3(gt, odL, qz, az);
                  j0 += Lz * A[comp][0][az];
                  j1 += Lz * A[comp][1][az];
                  j2 += dLz * A[comp][2][az];
                  vv += Lz * A[comp][3][az];
               }
               J[comp][0] = j0; J[comp][1] = j1; J[comp][2] = j2; v[comp] = vv;
            }
            const double w3 = wxy * T1(gt, oW, qz);
            const double aw = a.alpha * w3;
            double detJ;
            if (RMH_ADJ_CROSS)
            {
               // adj(J) v and det J through two cross products (c_k = dX/dxi_k, the columns of J; the rows of adj(J) are
               // c1 x c2, c2 x c0, c0 x c1 -- remhos_lo.cpp:1168-1180 -- so (adj(J) v)_0 = v.(c1 x c2) = -c1.(v x c2),
               // (adj(J) v)_1 = c0.(v x c2), (adj(J) v)_2 = v.(c0 x c1), det J = c2.(c0 x c1)): 24 instead of 30 operations
               // per quadrature point; p = 3 only (+1.0 %; at p = 6 the different register lifetimes cost 8 %)
               const double n0 = J[1][0] * J[2][1] - J[2][0] * J[1][1];
               const double n1 = J[2][0] * J[0][1] - J[0][0] * J[2][1];
               const double n2 = J[0][0] * J[1][1] - J[1][0] * J[0][1];
               const double c0 = v[1] * J[2][2] - v[2] * J[1][2];
               const double c1 = v[2] * J[0][2] - v[0] * J[2][2];
               const double c2 = v[0] * J[1][2] - v[1] * J[0][2];
               detJ = J[0][2] * n0 + J[1][2] * n1 + J[2][2] * n2;
               Dq[0][qz] = -aw * (J[0][1] * c0 + J[1][1] * c1 + J[2][1] * c2);
               Dq[1][qz] = aw * (J[0][0] * c0 + J[1][0] * c1 + J[2][0] * c2);
               Dq[2][qz] = aw * (v[0] * n0 + v[1] * n1 + v[2] * n2);
            }
            else
            {
               // adj(J), rows as in remhos_lo.cpp:1168-1180
               const double A11 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
               const double A12 = J[2][1] * J[0][2] - J[0][1] * J[2][2];
               const double A13 = J[0][1] * J[1][2] - J[1][1] * J[0][2];
               const double A21 = J[2][0] * J[1][2] - J[1][0] * J[2][2];
               const double A22 = J[0][0] * J[2][2] - J[0][2] * J[2][0];
               const double A23 = J[1][0] * J[0][2] - J[0][0] * J[1][2];
               const double A31 = J[1][0] * J[2][1] - J[2][0] * J[1][1];
               const double A32 = J[2][0] * J[0][1] - J[0][0] * J[2][1];
               const double A33 = J[0][0] * J[1][1] - J[0][1] * J[1][0];
               detJ = J[0][0] * A11 + J[0][1] * A21 + J[0][2] * A31;
               Dq[0][qz] = aw * (A11 * v[0] + A12 * v[1] + A13 * v[2]);
               Dq[1][qz] = aw * (A21 * v[0] + A22 * v[1] + A23 * v[2]);
               Dq[2][qz] = aw * (A31 * v[0] + A32 * v[1] + A33 * v[2]);
            }
            wl[qz] = w3 * detJ;
         }
      }
      // w detJ is needed again by the mass applies of the PCG: whole columns keep it in registers, split columns in LDS
      // (in registers, any form of it costs the kernel ~60 VGPRs of spills around the column phase)
#pragma unroll
      for (int qz = 0; qz < NQ; qz++)
      {
         if (ZS) { if (HAS_HO) { s_wdl[(cc - 64) * Q + zo1 + qz] = wl[qz]; } }
         else { wd[qz] = wl[qz]; }
      }
      // pass 2: grad u, D.grad u and the z-leg of the three test contractions
      const int u1o = (int)(RMH_W(ceb) - lds) + oU1 + qx * S2;
      const double *U1b = OPQ ? lds + opaque_lds_offset(u1o) : lds + u1o;                // B.u row of qx
      const double *U1g = OPQ ? lds + opaque_lds_offset(u1o + Q * S2) : lds + u1o + Q * S2; // G.u row of qx
      const double *tBy = OPQ ? lds + opaque_lds_offset((int)(stab - lds) + oB + qy * D) : stab + oB + qy * D;
      const double *tGy = OPQ ? lds + opaque_lds_offset((int)(stab - lds) + oG + qy * D) : stab + oG + qy * D;
      double By[D], Gy[D]; // (rows of qy for the u contractions: read behind the geometry pass, whose registers they would take)
#pragma unroll
      for (int k = 0; k < D; k++)
      {
         By[k] = tBy[k];
         Gy[k] = tGy[k];
      }
      double UB[D], UG[D], UU[D];
#pragma unroll
      for (int iz = 0; iz < D; iz++)
      {
         double b0 = 0, b1 = 0, b2 = 0;
#pragma unroll
         for (int iy = 0; iy < D; iy++)
         {
            const double ub = U1b[iy + D * iz];
            const double ug = U1g[iy + D * iz];
            b0 += By[iy] * ug;
            b1 += Gy[iy] * ub;
            b2 += By[iy] * ub;
         }
         UB[iz] = b0; UG[iz] = b1; UU[iz] = b2;
      }
#pragma unroll
      for (int qz = 0; qz < NQ; qz++)
      {
         if (ZS) { sched_fence(); }
         tabp gt = RMH_TABK();
         double gx = 0, gy = 0, gz = 0;
#pragma unroll
         for (int iz = 0; iz < D; iz++)
         {
            const double Bz = TD(gt, oB, qz, iz), Gz = TD(gt, oG, qz, iz);
            gx += Bz * UB[iz];
            gy += Bz * UG[iz];
            gz += Gz * UU[iz];
         }
         const double g = Dq[0][qz] * gx + Dq[1][qz] * gy + Dq[2][qz] * gz;
         const double wdq = wl[qz];
         if (ZS) { sched_fence(); }
         // test along z: r0: GL nodal basis x (D.grad u); r1: Bernstein x w detJ (lumped mass);
         //               r2: GL basis squared x w detJ (Jacobi diagonal)
#pragma unroll
         for (int iz = 0; iz < D; iz++)
         {
            r0[iz] += TD(gt, HAS_HO ? oBg : oB, qz, iz) * g;
            r1[iz] += TD(gt, oB, qz, iz) * wdq;
            if (HAS_HO) { r2[iz] += TD(gt, oBg2, qz, iz) * wdq; }
         }
      }
   };
   if (col)
   {
      if (zsplit) { column_pass(std::integral_constant<bool, CSPL>()); }
      else { column_pass(std::false_type()); }
   }
   RMH_STAMP(31);
   if (zsplit)
   {
      // the three thirds of a column's z-leg sums, added in the order of qz (DPP row_shl:1 and row_shl:2: the values of
      // the next two lanes of the row; every lane of the wavefront takes part)
#pragma unroll
      for (int iz = 0; iz < D; iz++)
      {
         r0[iz] = (r0[iz] + dpp_value<0x101>(r0[iz])) + dpp_value<0x102>(r0[iz]);
         r1[iz] = (r1[iz] + dpp_value<0x101>(r1[iz])) + dpp_value<0x102>(r1[iz]);
         if (HAS_HO) { r2[iz] = (r2[iz] + dpp_value<0x101>(r2[iz])) + dpp_value<0x102>(r2[iz]); }
      }
   }
   if (LO4)
   {
      __syncthreads(); // the face rows are complete
      lumped_face_products();
      __syncthreads();
      lumped_face_gather();
   }
   __syncthreads(); // R3 overlays the phase A-C data: every thread is done with nodes, u, traces, U1
   if (col)
   {
      double *R3 = RMH_W(ceb) + oR3;
      if (zt == 0)
      {
#pragma unroll
         for (int iz = 0; iz < D; iz++)
         {
            R3[0 * C::R3S + qy * C::QYS + qx * D + iz] = r0[iz];
            R3[1 * C::R3S + qy * C::QYS + qx * D + iz] = r1[iz];
            R3[2 * C::R3S + qy * C::QYS + qx * D + iz] = r2[iz];
         }
      }
      // (read here, not with the other rows of qy: it is first used by the mass apply and would only occupy registers
      // through the column phase)
#pragma unroll
      for (int k = 0; k < D; k++) { Bgy[k] = stab[oBg + qy * D + k]; }
   }
   __syncthreads();

   RMH_STAMP(3);
#if defined(RMH_NOP_PROBE) && defined(__HIP_DEVICE_COMPILE__)
   // diagnostic build only (tools/experiments/r06_jobs/nop_probe.sh): RMH_NOP_PROBE extra VALU issue slots per wavefront in an
   // issue-bound phase -- how much a stage kernel's time responds to its VALU instruction count, order by order
#pragma unroll
   for (int i_ = 0; i_ < RMH_NOP_PROBE; i_++) { asm volatile("v_nop"); }
#endif
   // ---- phase F: y-leg of the three test contractions (R2 overlays U1) ---------------------------------
   // (split workgroups: the three tensors are divided between the wavefronts -- r = 0, 1 / r = 2 -- not the outputs of
   // a line: in place, a line's outputs overwrite its own inputs)
   // (several elements per workgroup: the Q D tasks of an element start at a multiple of 16 lanes where the workgroup has
   // the threads -- p = 3: 24 tasks in 32 lanes -- so that the 16-lane groups of the LDS accesses do not straddle two
   // element blocks, whose lines would share banks)
   constexpr int TPE = (NB > 1 && NB * ((Q * D + 15) / 16 * 16) <= NT) ? (Q * D + 15) / 16 * 16 : Q * D;
   for (int k = ptid; k < NB * TPE; k += PNT)
   {
      const int eb = k / TPE, rem = k % TPE;
      if (rem >= Q * D) { continue; }
      const int q = rem / D, iz = rem % D;
      split_outputs<SPL, C::NR>(wv, [&](auto rlo, auto rhi) {
      // (PRE: the inputs of ALL the tensors of this task are read before the first output is stored.  The compiler cannot tell
      // that a tensor's stores leave the next tensor's line alone, and with the loads behind the stores the three tensors were
      // three dependent LDS round trips per task; one-element workgroups only -- p <= 3 has no registers to spare here)
      double inall[PRE ? C::NR : 1][Q];
      if (PRE)
      {
#pragma unroll
         for (int r = rlo; r < rhi; r++)
         {
            const double *R3 = RMH_W(eb) + oR3 + r * C::R3S + q * D + iz;
#pragma unroll
            for (int jy = 0; jy < Q; jy++) { inall[PRE ? r : 0][jy] = R3[jy * C::QYS]; }
         }
      }
#pragma unroll
      for (int r = rlo; r < rhi; r++)
      {
         const double *R3 = RMH_W(eb) + oR3 + r * C::R3S + q * D + iz;
         double in[Q];
#pragma unroll
         for (int jy = 0; jy < Q; jy++) { in[jy] = PRE ? inall[PRE ? r : 0][jy] : R3[jy * C::QYS]; }
         // [r][qx][iy + D*iz]; in place: [r][qx + Q*iy][iz], the first D entries of the line just read
         double *dst = C::INPLACE_Y ? RMH_W(eb) + oR3 + r * C::R3S + q * D + iz : RMH_W(eb) + oR2 + (r * Q + q) * C::R2S + D * iz;
         constexpr int dstr = C::INPLACE_Y ? C::QYS : 1;
         // p = 3 (round 6): the three tables of this leg -- 3 x 24 doubles = 144 scalar registers -- do not fit beside the kernel's
         // long-lived scalars, which the compiler parked in VGPR lanes around the phase (22 v_writelane + 22 v_readlane per
         // wavefront: VALU instructions in an issue-bound phase).  One opaque view per tensor keeps 48 table registers live at a
         // time: no scalar spills left in the kernel (with the x-pencils below: 2786 -> 2708 static VALU instructions, p = 3
         // +1.2 %, lo 4 at p = 3 +1.9 %, p = 2 +-0: only p = 3; bit-identical)
#ifndef RMH_YLEG_VIEW3
#define RMH_YLEG_VIEW3 (P == 3)
#endif
         constexpr bool YV3 = RMH_YLEG_VIEW3 && P < RMH_VIEW_MINP;
         std::conditional_t<YV3, tabp_const, tabp> gt = (std::conditional_t<YV3, tabp_const, tabp>)gtb;
         if constexpr (YV3) { gt = tab_view_c<P>(); } // (one view per tensor)
#pragma unroll
         for (int iy = 0; iy < D; iy++)
         {
            if constexpr (!YV3) { if (((iy) - (0)) % GQ == 0) { gt = RMH_TABK(); } } // (one view per group of outputs: GD)
            double acc = 0.0;
#pragma unroll
            for (int jy = 0; jy < Q; jy++)
            {
               // (TT: the output's table column from the transposed copy -- one wide scalar load instead of Q gathered entries)
               const double w = TT ? ((r == 0) ? gt[(HAS_HO ? C::oBgT : C::oBT) + iy * Q + jy] : (r == 2 ? gt[C::oBg2T + iy * Q + jy] : gt[C::oBT + iy * Q + jy]))
                                   : ((r == 0) ? gt[(HAS_HO ? oBg : oB) + jy * D + iy] : (r == 2 ? gt[oBg2 + jy * D + iy] : gt[oB + jy * D + iy]));
               acc += w * in[jy];
            }
            dst[iy * dstr] = acc;
         }
      }
      });
   }
   if (HAS_HO)
   {
      // the face rows are contracted along q1 here, in place (column (f, k2) of the face buffer belongs to one
      // thread: Q values in, D values out): F[(f*Q + k1)*D + k2] = sum_q1 Bg[q1][k1] F[(f*Q + q1)*D + k2], so that a
      // dof of phase G adds one value per face instead of a Q-term sum
      for (int k = NT - 1 - tid; k < NB * 6 * D; k += NT) // (from the top: the y-leg tasks occupy the low threads)
      {
         const int eb = k / (6 * D), rem = k % (6 * D);
         const int f = rem / D, k2 = rem % D;
         double *F = RMH_W(eb) + oF + f * Q * D + k2;
         double in[Q];
#pragma unroll
         for (int q1 = 0; q1 < Q; q1++) { in[q1] = F[q1 * D]; }
tabp gt = gtb;
#pragma unroll
         for (int k1 = 0; k1 < D; k1++)
         {
            if (((k1) - (0)) % GQ == 0) { gt = RMH_TABK(); } // (one view per group of outputs: GD)
            double acc = 0.0;
#pragma unroll
            for (int q1 = 0; q1 < Q; q1++) { acc += (TT ? gt[C::oBgT + k1 * Q + q1] : gt[oBg + q1 * D + k1]) * in[q1]; }
            F[k1 * D] = acc;
         }
      }
   }
   __syncthreads();

   RMH_STAMP(4);
   // ---- phase G: dof threads: x-leg, face contributions ------------------------------------------------
   // (RD solver: the element's u in the dof role is needed right after this phase; the reload -- an L2 hit -- is
   // issued here so that the x-leg hides it)
   double uu4[DR];
#pragma unroll
   for (int r = 0; r < DR; r++)
   {
      const int t = tid + r * NT;
      uu4[r] = 0.0;
      if (LO4 && t < NB * D3) { uu4[r] = a.u[(size_t)min(e0 + t / D3, a.e_end - 1) * D3 + t % D3]; }
   }
   // One element per workgroup, HO / lo 5 kernels (p >= 4): the x-leg as pencil tasks, in place -- a thread reads the Q
   // inputs of a line (iy, iz) of one tensor once and writes its D outputs over the first D of them, table rows as scalar
   // operands; the dof threads then pick up one value per tensor.  In the dof form every dof read Q inputs and Q table
   // entries from LDS per tensor (p = 6: 162 LDS reads per thread, now 27 + 9).  Split workgroups divide the tensors, as in
   // the y-leg.  Same sums in the same order.
   constexpr bool PX = NB == 1 && C::INPLACE_Y && !LO4 && !(RMH_CBG_REG);
   if (PX)
   {
      constexpr int rs2 = C::R3S;
      for (int k = ptid; k < D2; k += PNT)
      {
         const int iy = k / D, iz = k % D;
         split_outputs<SPL, C::NR>(wv, [&](auto rlo, auto rhi) {
            double inx[PRE ? C::NR : 1][Q]; // (PRE: every tensor's line is read before the first output is stored, see the y-leg)
            auto xleg_outputs = [&](const int r, double *line, const double (&in)[Q]) {
               tabp gt = gtb;
#pragma unroll
               for (int ix = 0; ix < D; ix++)
               {
                  if (ix % GQ == 0) { gt = RMH_TABK(); } // (one view per group of outputs: GD)
                  double acc = 0.0;
#pragma unroll
                  for (int jx = 0; jx < Q; jx++)
                  {
                     const double w = TT ? ((r == 0) ? gt[C::oBgT + ix * Q + jx] : (r == 2 ? gt[C::oBg2T + ix * Q + jx] : gt[C::oBT + ix * Q + jx]))
                                         : ((r == 0) ? gt[oBg + jx * D + ix] : (r == 2 ? gt[oBg2 + jx * D + ix] : gt[oB + jx * D + ix]));
                     acc += w * in[jx];
                  }
                  line[ix * D] = acc;
               }
            };
#pragma unroll
            for (int r = rlo; r < rhi; r++)
            {
               double *line = RMH_W(0) + oR2 + r * rs2 + iy * C::QYS + iz;
#pragma unroll
               for (int jx = 0; jx < Q; jx++) { inx[PRE ? r : 0][jx] = line[jx * D]; }
               if (!PRE) { xleg_outputs(r, line, inx[0]); }
            }
            if (PRE)
            {
#pragma unroll
               for (int r = rlo; r < rhi; r++) { xleg_outputs(r, RMH_W(0) + oR2 + r * rs2 + iy * C::QYS + iz, inx[PRE ? r : 0]); }
            }
         });
      }
      __syncthreads();
   }
   double rg[DR], mm[DR], dg[DR], zb[DR];
   // column ix of the GL basis table of each dof of this thread (x-legs): kept in registers through the PCG loop where
   // that is cheap (p <= 4); at p = 6 the 27 doubles starve the loop of registers -- every LDS read then reuses one
   // temporary and waits for it (read, wait, two FMAs, read, ...) -- and the x-back leg reads its row from the table
   constexpr bool CBG_REG = RMH_CBG_REG;
   double cBg[CBG_REG ? DR : 1][Q];
#pragma unroll
   for (int r = 0; r < DR; r++)
   {
      const int t = tid + r * NT;
      rg[r] = 0.0; mm[r] = 1.0; dg[r] = 1.0; zb[r] = 0.0;
      if (CBG_REG)
      {
#pragma unroll
         for (int jx = 0; jx < Q; jx++) { cBg[CBG_REG ? r : 0][jx] = 0.0; }
      }
      if (t < NB * D3)
      {
         const int eb = t / D3, i = t % D3;
         const int ix = i % D, i2 = i / D;
         const int idx[3] = {ix, i2 % D, i2 / D};
         const double *R2 = RMH_W(eb) + oR2;
         constexpr bool XB = BOTH && RMH_RD_XLEG;
         double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
         if (PX)
         {
            const int o2 = ix * D + idx[1] * C::QYS + idx[2];
            a0 = R2[0 * C::R3S + o2];
            a1 = R2[1 * C::R3S + o2];
            a2 = R2[2 * C::R3S + o2];
         }
#pragma unroll
         for (int jx = 0; jx < (PX ? 0 : Q); jx++)
         {
            const double bgx = stab[(HAS_HO ? oBg : oB) + jx * D + ix];
            if (CBG_REG) { cBg[CBG_REG ? r : 0][jx] = bgx; }
            const double bx = stab[oB + jx * D + ix];
            // R2[r][jx][iy + D*iz] -- in place it sits at [r][jx + Q*iy][iz] of R3
            const int o2 = C::INPLACE_Y ? jx * D + idx[1] * C::QYS + idx[2] : jx * C::R2S + i2;
            constexpr int rs2 = C::INPLACE_Y ? C::R3S : Q * C::R2S;
            const double r0x = R2[0 * rs2 + o2];
            a0 += bgx * r0x;
            if (XB) { a3 += bx * r0x; } // the RD solver's z, Bernstein-tested along x (y and z follow below)
            a1 += bx * R2[1 * rs2 + o2];
            if (HAS_HO) { a2 += stab[oBg2 + jx * D + ix] * R2[2 * rs2 + o2]; }
         }
         const double a0vol = XB ? a3 : a0;
         // faces: the GL nodal basis does not vanish on the faces, every dof sees all six
#pragma unroll
         for (int c = 0; c < (HAS_HO ? 3 : 0); c++)
         {
            const int c1 = (c + 1) % 3, c2 = (c + 2) % 3;
            const int kc = idx[c], k1 = idx[c1], k2 = idx[c2];
#pragma unroll
            for (int side = 0; side < 2; side++)
            {
               // (face rows already contracted along q1 in phase F)
               a0 += stab[C::oBgE + side * D + kc] * RMH_W(eb)[oF + ((2 * c + side) * Q + k1) * D + k2];
            }
         }
         rg[r] = a0; mm[r] = a1; dg[r] = a2;
         zb[r] = BOTH ? a0vol : a0; // z = K_vol u: Bernstein-tested in the RD-only kernel, GL-tested (converted below) otherwise
      }
   }

   double dlo[DR]; // du_LO of the RD solver (MODE 2 / 3)
   RMH_STAMP(27);
#pragma unroll
   for (int r = 0; r < DR; r++) { dlo[r] = 0.0; }
   if (LO4)
   {
      // ---- residual distribution per element (remhos_lo.cpp:1702-1800); zb = z = K_vol u, mm = lumped mass
      constexpr int NS = C::NS;
      const double eps = 1.E-15, gamma = 1.0;
      double t0[DR], t1[DR], t2[DR], xSum[DR], rhoP[DR], rhoN[DR];
      int ring4 = 0;
      constexpr bool WD = RMH_WAVE_DOT && C::WAVE_ALIGNED && DR == 2; // element sums stay in the wavefront (see wave_bcast)
      __syncthreads(); // R2 has been consumed: the front of W is free
      if (BOTH)
      {
         // z (GL-tested along y and z; the x-leg above tested it with the Bernstein basis) -> z (Bernstein-tested):
         // Cf along y and along z through the sA / sB slots
#pragma unroll
         for (int r = 0; r < DR; r++)
         {
            const int t = tid + r * NT;
            if (t < NB * D3) { RMH_W(t / D3)[oSA + t % D3] = zb[r]; }
         }
         __syncthreads();
         constexpr int dir0 = RMH_RD_XLEG ? 1 : 0;
         for (int dir = dir0; dir < 3; dir++)
         {
            const int oin = ((dir - dir0) & 1) ? oSB : oSA, oout = ((dir - dir0) & 1) ? oSA : oSB;
            basis_leg(IntC<C::oCf>{}, dir, dir == dir0, oin, oout);
            __syncthreads();
            if (dir == 2)
            {
#pragma unroll
               for (int r = 0; r < DR; r++)
               {
                  const int t = tid + r * NT;
                  if (t < NB * D3) { zb[r] = RMH_W(t / D3)[oout + t % D3]; }
               }
            }
         }
         // (the dof threads' reads of the result precede the next stores into these slots: the PCG prelude is a barrier away)
         if (!WD) { __syncthreads(); }
      }
#pragma unroll
      for (int r = 0; r < DR; r++)
      {
         const int t = tid + r * NT;
         if (!WD && t < NB * D3) { RMH_W(t / D3)[oSA + t % D3] = uu4[r]; }
         t0[r] = uu4[r];
         t1[r] = fmax(0., zb[r]);
         t2[r] = fmin(0., zb[r]);
      }
      RMH_STAMP(28);
      batch_dot<C>(tid, t0, xSum, lds, s_acc, ring4);
      batch_dot2<C>(tid, t1, t2, rhoP, rhoN, lds, s_acc, ring4);
      RMH_STAMP(29);
      // element extrema (el[0..1]) and the sums of the subcell fluctuations (8 partial sums each, el[2..17])
      double elo[DR], ehi[DR], sfP[DR], sfN[DR];
#pragma unroll
      for (int r = 0; r < DR; r++) { elo[r] = 0; ehi[r] = 0; sfP[r] = 0; sfN[r] = 0; }
      if (WD)
      {
         // p = 3: a wavefront holds the whole element in each of its dof rounds -- the sums of the subcell fluctuations are
         // formed in the dof role (dof i < NS reads subcell i) like the other element sums, and the extrema with one packed
         // reduction (see the extrema of the new state in phase K); everything stays in the wavefront: no LDS, no barrier
         double fp[DR], fn[DR];
#pragma unroll
         for (int r = 0; r < DR; r++)
         {
            const int t = tid + r * NT;
            const double f = (t < NB * D3 && t % D3 < NS) ? (RMH_W(t / D3) + C::oSub)[t % D3] : 0.0;
            fp[r] = fmax(0., f);
            fn[r] = fmin(0., f);
         }
         batch_dot2<C>(tid, fp, fn, sfP, sfN, lds, s_acc, ring4);
         const bool has1 = tid + NT < NB * D3;
         double y0 = uu4[0], y1 = has1 ? uu4[DR == 2 ? 1 : 0] : INFINITY;
         double z0 = -uu4[0], z1 = has1 ? -uu4[DR == 2 ? 1 : 0] : INFINITY;
         swap32(y0, y1);
         swap32(z0, z1);
         double mn = fmin(y0, y1), nx = fmin(z0, z1);
         swap16(mn, nx);
         double x = fmin(mn, nx); // rows: {min round 0, -max round 0, min round 1, -max round 1}
         x = dpp_minmax<0xB1, 0xF, true>(x);
         x = dpp_minmax<0x4E, 0xF, true>(x);
         x = dpp_minmax<0x141, 0xF, true>(x);
         x = dpp_minmax<0x140, 0xF, true>(x);
         elo[0] = wave_bcast<15>(x);
         ehi[0] = -wave_bcast<31>(x);
         elo[DR == 2 ? 1 : 0] = wave_bcast<47>(x);
         ehi[DR == 2 ? 1 : 0] = -wave_bcast<63>(x);
         if (!BOTH)
         {
#pragma unroll
            for (int r = 0; r < DR; r++)
            {
               const int t = tid + r * NT;
               if ((tid & 63) == 0 && t < NB * D3 && e0 + t / D3 < a.e_end)
               {
                  a.xe_min[e0 + t / D3] = elo[r];
                  a.xe_max[e0 + t / D3] = ehi[r];
               }
            }
         }
      }
      else if (C::WAVE_ALIGNED)
      {
#pragma unroll
         for (int r = 0; r < DR; r++)
         {
            const int t = tid + r * NT;
            const double lo = wave_minmax<true>(t < NB * D3 ? uu4[r] : INFINITY);
            const double hi = wave_minmax<false>(t < NB * D3 ? uu4[r] : -INFINITY);
            if ((tid & 63) == 63 && t < NB * D3)
            {
               double *el = RMH_W(t / D3) + oM1;
               el[0] = lo; el[1] = hi;
               if (!BOTH && e0 + t / D3 < a.e_end)
               {
                  a.xe_min[e0 + t / D3] = lo;
                  a.xe_max[e0 + t / D3] = hi;
               }
            }
         }
      }
      else if (NB == 1)
      {
         // one element per workgroup (p = 4, 5, 6): thread, wavefront (DPP), one slot of four doubles per wavefront --
         // {min, max, sum of fluct^+, sum of fluct^-}; the readers combine the wavefronts' slots in order (one thread
         // looping over the D^3 values and eight over the subcells were 10 % of the lo 4 stage at p = 6)
         double lo = INFINITY, hi = -INFINITY, sp = 0.0, sn = 0.0;
#pragma unroll
         for (int r = 0; r < DR; r++)
         {
            const bool in = tid + r * NT < D3;
            lo = fmin(lo, in ? uu4[r] : INFINITY);
            hi = fmax(hi, in ? uu4[r] : -INFINITY);
         }
         const double *fl = RMH_W(0) + C::oSub;
#pragma unroll
         for (int r = 0; r < (NS + NT - 1) / NT; r++)
         {
            const int m = tid + r * NT;
            const double f = m < NS ? fl[m] : 0.0;
            sp += fmax(0., f);
            sn += fmin(0., f);
         }
         constexpr int NW = NT / 64;
         if (RMH_PACKED_SUMS)
         {
            // (two packed reductions instead of four: lane 31 holds the minimum and the sum of fluct^+, lane 63 -max and the sum of fluct^-)
            const double mm = wave_min_negmax(lo, hi), ss = wave_sum2(sp, sn);
            double *el = RMH_W(0) + oM1 + 4 * (tid >> 6);
            if ((tid & 63) == 31) { el[0] = mm; el[2] = ss; }
            if ((tid & 63) == 63) { el[1] = -mm; el[3] = ss; }
         }
         else
         {
            lo = wave_minmax<true>(lo);
            hi = wave_minmax<false>(hi);
            sp = wave_sum(sp);
            sn = wave_sum(sn);
            if ((tid & 63) == 63)
            {
               double *el = RMH_W(0) + oM1 + 4 * (tid >> 6);
               el[0] = lo; el[1] = hi; el[2] = sp; el[3] = sn;
            }
         }
      }
      else if (tid < NB)
      {
         const double *uu = RMH_W(tid) + oSA;
         double lo = INFINITY, hi = -INFINITY;
         for (int i = 0; i < D3; i++)
         {
            lo = fmin(lo, uu[i]);
            hi = fmax(hi, uu[i]);
         }
         double *el = RMH_W(tid) + oM1;
         el[0] = lo; el[1] = hi;
         if (!BOTH && e0 + tid < a.e_end)
         {
            a.xe_min[e0 + tid] = lo;
            a.xe_max[e0 + tid] = hi;
         }
      }
      if (!WD && (NB != 1 || C::WAVE_ALIGNED))
      {
         constexpr int CH = 8, CL = (NS + CH - 1) / CH;
         for (int k = tid; k < NB * CH; k += NT)
         {
            const double *fl = RMH_W(k / CH) + C::oSub;
            const int m0 = (k % CH) * CL;
            double sp = 0.0, sn = 0.0;
            for (int m = m0; m < m0 + CL && m < NS; m++)
            {
               sp += fmax(0., fl[m]);
               sn += fmin(0., fl[m]);
            }
            double *el = RMH_W(k / CH) + oM1;
            el[2 + k % CH] = sp;
            el[10 + k % CH] = sn;
         }
      }
      if (!WD) { __syncthreads(); }
      RMH_STAMP(30);
      double rd_auxP = 0, rd_invP = 0, rd_auxN = 0, rd_invN = 0, rd_rWP = 0, rd_rWN = 0; // (element scalars of the weights, see UNI below)
#pragma unroll
      for (int r = 0; r < DR; r++)
      {
         const int t = tid + r * NT;
         if (t < NB * D3)
         {
            const int eb = t / D3, i = t % D3;
            const int ix = i % D, iy = (i / D) % D, iz = i / D2;
            const double *el = RMH_W(eb) + oM1, *fl = RMH_W(eb) + C::oSub;
            double xe_min = WD ? elo[r] : el[0], xe_max = WD ? ehi[r] : el[1];
            double sumFluctP = sfP[r], sumFluctN = sfN[r];
            if (WD) {}
            else if (NB == 1 && !C::WAVE_ALIGNED)
            {
               sumFluctP = el[2];
               sumFluctN = el[3];
#pragma unroll
               for (int w = 1; w < NT / 64; w++)
               {
                  xe_min = fmin(xe_min, el[4 * w]);
                  xe_max = fmax(xe_max, el[4 * w + 1]);
                  sumFluctP += el[4 * w + 2];
                  sumFluctN += el[4 * w + 3];
               }
               if (!BOTH && t == 0 && e0 < a.e_end)
               {
                  a.xe_min[e0] = xe_min;
                  a.xe_max[e0] = xe_max;
               }
            }
            else
            {
#pragma unroll
               for (int c = 0; c < 8; c++)
               {
                  sumFluctP += el[2 + c];
                  sumFluctN += el[10 + c];
               }
            }
            const double ui = uu4[r];
            double nwP = 0.0, nwN = 0.0;
            // the (up to) eight subcells of this dof, without branches: a subcell outside the element is read at a
            // clamped index with ratio 0; a ratio > 0 belongs to eq. (58) (subcell maximum), one < 0 -- stored negated
            // -- to eq. (59) (subcell minimum)
#pragma unroll
            for (int dz = 1; dz >= 0; dz--)
            {
#pragma unroll
               for (int dy = 1; dy >= 0; dy--)
               {
#pragma unroll
                  for (int dx = 1; dx >= 0; dx--)
                  {
                     const int mx = ix - dx, my = iy - dy, mz = iz - dz;
                     const bool in = mx >= 0 && mx < P && my >= 0 && my < P && mz >= 0 && mz < P;
                     const int m = min(max(mx, 0), P - 1) + P * (min(max(my, 0), P - 1) + P * min(max(mz, 0), P - 1));
                     const double ratio = in ? fl[2 * NS + m] : 0.0;
                     const double d = fl[1 * NS + m] - ui;
                     nwP += fmax(ratio, 0.) * d; // eq. (58)
                     nwN -= fmin(ratio, 0.) * d; // eq. (59)
                  }
               }
            }
            const double sumWeightsP = D3 * xe_max - xSum[r] + eps;
            const double sumWeightsN = D3 * xe_min - xSum[r] - eps;
            // (UNI: four of the seven divisions of a dof have element sums on both sides, two more an element sum below the line --
            // formed, or their reciprocal refined, once per lane instead of once per dof round; the same bits)
            if (!UNI || r == 0)
            {
               rd_auxP = fdiv(gamma, rhoP[r] + eps);
               rd_invP = fdiv(1., sumFluctP + eps);
               rd_auxN = fdiv(gamma, rhoN[r] - eps);
               rd_invN = fdiv(1., sumFluctN - eps);
               if (UNI) { rd_rWP = fdiv_rcp(sumWeightsP); rd_rWN = fdiv_rcp(sumWeightsN); }
            }
            double weightP = UNI ? fdiv_by(xe_max - ui, sumWeightsP, rd_rWP) : fdiv(xe_max - ui, sumWeightsP);
            double weightN = UNI ? fdiv_by(xe_min - ui, sumWeightsN, rd_rWN) : fdiv(xe_min - ui, sumWeightsN);
            double aux = rd_auxP;
            weightP *= 1. - fmin(aux * sumFluctP, 1.);
            weightP += fmin(aux, rd_invP) * nwP;
            aux = rd_auxN;
            weightN *= 1. - fmin(aux * sumFluctN, 1.);
            weightN += fmax(aux, rd_invN) * nwN;
            const double duf = RMH_W(eb)[C::oDuf + i];
            dlo[r] = fdiv(duf + weightP * rhoP[r] + weightN * rhoN[r], mm[r]);
            if (!BOTH && e0 + eb < a.e_end)
            {
               a.du[(size_t)e0 * D3 + t] = dlo[r];
               a.m[(size_t)e0 * D3 + t] = mm[r];
            }
         }
      }
      if (!BOTH)
      {
         RMH_STAMP(7);
         RMH_STAMP_FLUSH();
         return;
      }
      __syncthreads(); // the PCG reuses the front of W
   }
   RMH_STAMP(5);
   // ---- phase I: element-local PCG in the GL nodal basis (DGMassInverse) ----------------------------------
   // generic orders: the deterministic reductions stage their operands in the sB slot of W, which may
   // overlap the tail of R2 that slower threads are still reading in phase G
   if (!C::WAVE_ALIGNED) { __syncthreads(); }
   // kernel arguments that are only needed from here on (limiter, RK update, stores) are read through a late view of
   // the kernarg segment: held in scalar registers from the kernel's first instruction they cost ~30 SGPRs through
   // phases A-J and push table values into VGPR-lane spills
   LateArgs L = late_args(a);
   // fused stage: the global reads of the limiter part are issued here so that they are in flight during
   // the PCG iterations (u is an L2 hit: this workgroup read it in phase A)
   double uu[DR], xb[DR], slo[NLS], shi[NLS];
   if (FUSED)
   {
      // (late arguments read ONCE and together, in front of the exec-masked blocks: inside them each was a scalar load with its
      // own exposed wait -- three dependent round trips here, one per dof round below; p = 5 +1.2 %, p = 3, 4, 6 +0.6 ... +0.8 %)
      int ne_ownL = L.ne_owned, gh_mstrL = L.gh_mstride, e_lastL = L.e_end - 1;
      const double *pminL = L.xe_min, *pmaxL = L.xe_max, *gminL = L.gh_min, *gmaxL = L.gh_max, *x_baseL = L.x_base;
#if defined(__HIP_DEVICE_COMPILE__)
      // (pinned here: the compiler otherwise sinks each load back into the block of its first use)
      asm volatile("" : "+s"(ne_ownL), "+s"(gh_mstrL), "+s"(e_lastL), "+s"(pminL), "+s"(pmaxL), "+s"(gminL), "+s"(gmaxL), "+s"(x_baseL));
#endif
#pragma unroll
      for (int j = 0; j < NLS; j++)
      {
         const int k = tid + j * NT;
         slo[j] = INFINITY; shi[j] = -INFINITY;
         const int nb = (k < NB * 27) ? s_sti[k] : -1;
         if (nb >= 0)
         {
            // (owned or ghost: the two base pointers are scalars, the select happens on the per-lane address)
            const bool own = nb < ne_ownL;
            const size_t off = own ? (size_t)nb : (size_t)(nb - ne_ownL) * gh_mstrL;
            slo[j] = (own ? pminL : gminL)[off];
            shi[j] = (own ? pmaxL : gmaxL)[off];
         }
      }
#pragma unroll
      for (int r = 0; r < DR; r++)
      {
         const int t = tid + r * NT;
         uu[r] = 0.0; xb[r] = 0.0;
         if (t < NB * D3)
         {
            const size_t g = (size_t)min(e0 + t / D3, e_lastL) * D3 + t % D3;
            uu[r] = BOTH ? uu4[r] : a.u[g]; // (HO + RD: already reloaded for the RD solver)
            if (x_baseL) { xb[r] = x_baseL[g]; }
         }
      }
   }
   double xg[DR], dd[DR], nom[DR], tol[DR], tmp[DR], red[DR];
   int its[DR];
   int ring = 0;
#pragma unroll
   for (int r = 0; r < DR; r++)
   {
      xg[r] = 0.0;
      dg[r] = fdiv(1.0, dg[r]); // from here on dg holds the inverse diagonal (one division per dof instead of two per iteration)
      dd[r] = rg[r] * dg[r];
      tmp[r] = rg[r] * dd[r];
      its[r] = 0;
   }
   // nom = r.z; kept for the constant-mode completion behind the back-transform: the sum of the right-hand side (the
   // element's exact mass rate 1^T b, both bases sum to one) and the element's volume 1^T M 1 = sum of the lumped mass
   batch_dot_keep2<C>(tid, tmp, rg, mm, nom, lds, s_acc, ring);
   bool act[DR];
   {
      bool any = false;
#pragma unroll
      for (int r = 0; r < DR; r++)
      {
         tol[r] = fmax(a.rel2 * nom[r], a.abs2);
         act[r] = (tid + r * NT < NB * D3) && (nom[r] > tol[r]);
         any = any || act[r];
      }
      if (any) { raise_flag(&s_flag[0]); } // visible after the first barrier of the loop
   }
   // One PCG iteration; false: the loop is left.  The FIRST iteration is peeled (RMH_PCG_PEEL): under the -pa rule the bench meshes
   // need one iteration at p = 3 -- as straight-line code it carries none of the register copies of a loop's back edge (the
   // "tail" of the loop held 20 v_mov_b64 + 9 v_mov_b32 per trip, the exit trip included).
#ifndef RMH_PCG_PEEL
#define RMH_PCG_PEEL (NB > 1)
#endif
   auto pcg_iteration = [&](const int it) -> bool {
      RMH_STAMP(9);
      // Ad = M_g d
#pragma unroll
      for (int r = 0; r < DR; r++)
      {
         const int t = tid + r * NT;
         if (t < NB * D3) { RMH_W(t / D3)[oSA + t % D3] = dd[r]; }
      }
      if (tid == 0) { s_flag[(it + 1) & 1] = 0; }
      __syncthreads();
      if (!s_flag[it & 1]) { return false; } // no element of the batch is active any more
      RMH_STAMP(10);
      for (int k = ptid; k < NB * D2; k += PNT)
      {
         const int eb = k / D2, i2 = k % D2;
         const double *src = RMH_W(eb) + oSA + D * i2;
         double in[D];
#pragma unroll
         for (int ix = 0; ix < D; ix++) { in[ix] = src[ix]; }
         double *dst = RMH_W(eb) + oM1 + i2;
         split_outputs<SPL, Q>(wv, [&](auto qlo, auto qhi) {
tabp gt = gtb;
#pragma unroll
            for (int q = qlo; q < qhi; q++)
            {
               if (((q) - (qlo)) % GD == 0) { gt = RMH_TABK(); } // (one view per group of outputs: GD)
               double acc = 0.0;
#pragma unroll
               for (int ix = 0; ix < D; ix++) { acc += gt[oBg + q * D + ix] * in[ix]; }
               dst[q * S2] = acc;
            }
         });
      }
      __syncthreads();
      RMH_STAMP(11);
      if (col && zt == 0) // (split columns: the first of the three lanes runs the whole column, w detJ from LDS)
      {
         double rz[D], wq[Q];
#pragma unroll
         for (int iz = 0; iz < D; iz++) { rz[iz] = 0.0; }
         if (zsplit)
         {
#pragma unroll
            for (int qz = 0; qz < Q; qz++) { wq[qz] = s_wdl[(cc - 64) * Q + qz]; }
         }
         else
         {
#pragma unroll
            for (int qz = 0; qz < Q; qz++) { wq[qz] = wd[qz]; }
         }
         const double *M1 = OPQ ? lds + opaque_lds_offset((int)(RMH_W(ceb) - lds) + oM1 + qx * S2) : RMH_W(ceb) + oM1 + qx * S2;
         double Y[D];
#pragma unroll
         for (int iz = 0; iz < D; iz++)
         {
            double acc = 0.0;
#pragma unroll
            for (int iy = 0; iy < D; iy++) { acc += Bgy[iy] * M1[iy + D * iz]; }
            Y[iz] = acc;
         }
tabp gt = gtb;
#pragma unroll
         for (int qz = 0; qz < Q; qz++)
         {
            if (((qz) - (0)) % GD == 0) { gt = RMH_TABK(); } // (one view per group of outputs: GD)
            double acc = 0.0;
#pragma unroll
            for (int iz = 0; iz < D; iz++) { acc += gt[oBg + qz * D + iz] * Y[iz]; }
            acc *= wq[qz];
#pragma unroll
            for (int iz = 0; iz < D; iz++) { rz[iz] += gt[oBg + qz * D + iz] * acc; }
         }
         double *R3 = RMH_W(ceb) + oR3c + cc * D;
#pragma unroll
         for (int iz = 0; iz < D; iz++) { R3[iz] = rz[iz]; }
      }
      __syncthreads();
      RMH_STAMP(12);
      static_assert(!SPL || NB * Q * D <= 64, "split phases: one task per lane");
      for (int k = ptid; k < NB * TPE; k += PNT)
      {
         const int eb = k / TPE, rem = k % TPE;
         if (rem >= Q * D) { continue; }
         const int q = rem / D, iz = rem % D;
         const double *R3 = RMH_W(eb) + oR3c + q * D + iz;
         double in[Q];
#pragma unroll
         for (int jy = 0; jy < Q; jy++) { in[jy] = R3[Q * jy * D]; }
         double *dst = RMH_W(eb) + oM1 + q * S2 + D * iz;
         split_outputs<SPL, D>(wv, [&](auto ylo, auto yhi) {
tabp gt = gtb;
#pragma unroll
            for (int iy = ylo; iy < yhi; iy++)
            {
               if (((iy) - (ylo)) % GQ == 0) { gt = RMH_TABK(); } // (one view per group of outputs: GD)
               double acc = 0.0;
#pragma unroll
               for (int jy = 0; jy < Q; jy++) { acc += (TT ? gt[C::oBgT + iy * Q + jy] : gt[oBg + jy * D + iy]) * in[jy]; }
               dst[iy] = acc;
            }
         });
      }
      __syncthreads();
      RMH_STAMP(13);
      double Ad[DR];
#pragma unroll
      for (int r = 0; r < DR; r++)
      {
         const int t = tid + r * NT;
         Ad[r] = 0.0;
         tmp[r] = 0.0;
         if (t < NB * D3)
         {
            const int eb = t / D3, i = t % D3;
            const int i2 = i / D;
            const double *R2 = RMH_W(eb) + oM1 + i2;
            const double *brow = stab + (HAS_HO ? oBg : oB) + i % D;
            double acc = 0.0;
#pragma unroll
            for (int jx = 0; jx < Q; jx++) { acc += (CBG_REG ? cBg[CBG_REG ? r : 0][jx] : brow[jx * D]) * R2[jx * S2]; }
            Ad[r] = acc;
            tmp[r] = dd[r] * acc;
         }
      }
      RMH_STAMP(8);
      batch_dot<C>(tid, tmp, red, lds, s_acc, ring); // den = d.Ad
      // (UNI -- one element per workgroup: every in-range round of every lane holds the same element scalars, round 0 is always in
      // range; the quotient is formed once instead of once per round behind its own branch: the same value, DR - 1 dependent
      // division chains less per use)
      const double al_uni = UNI ? fdiv(nom[0], red[0]) : 0.0;
#pragma unroll
      for (int r = 0; r < DR; r++)
      {
         const bool ok = act[r] && red[r] > 0.0;
         const double al = ok ? (UNI ? al_uni : fdiv(nom[r], red[r])) : 0.0;
         if (act[r] && !ok) { tol[r] = INFINITY; } // breakdown: freeze this element
         xg[r] += al * dd[r];
         rg[r] -= al * Ad[r];
         tmp[r] = rg[r] * (rg[r] * dg[r]);
      }
      RMH_STAMP(14);
      // capped solve (rmh_set_mass_tol with max_iter applies): the last admitted iteration ends with the update of x and
      // r -- its r.z would only decide about an iteration that is not going to happen
      if (it == a.max_iter - 1)
      {
#pragma unroll
         for (int r = 0; r < DR; r++) { its[r] += act[r] ? 1 : 0; }
         return false;
      }
      batch_dot<C>(tid, tmp, red, lds, s_acc, ring); // betanom = r.z
      RMH_STAMP(15);
      bool any = false;
      // (the new direction only where the element goes on: a converged element -- under the -pa rule every element of the bench meshes
      // after its first iteration -- skips the quotient, a nine-instruction chain per round; its d is not read again)
      double be_uni = 0.0;
      if (UNI && act[0] && red[0] > tol[0]) { be_uni = fdiv(red[0], nom[0]); }
#pragma unroll
      for (int r = 0; r < DR; r++)
      {
         if (act[r])
         {
            if (red[r] > tol[r]) { dd[r] = rg[r] * dg[r] + (UNI ? be_uni : fdiv(red[r], nom[r])) * dd[r]; }
            nom[r] = red[r];
            its[r]++;
         }
         act[r] = act[r] && (nom[r] > tol[r]);
         any = any || act[r];
      }
      if (any) { raise_flag(&s_flag[(it + 1) & 1]); } // read after the first barrier of the next iteration
      return true;
   };
   if (RMH_PCG_PEEL)
   {
      if (a.max_iter > 0 && pcg_iteration(0))
      {
         for (int it = 1; it < a.max_iter; it++) { if (!pcg_iteration(it)) { break; } }
      }
   }
   else
   {
      for (int it = 0; it < a.max_iter; it++) { if (!pcg_iteration(it)) { break; } }
   }

   RMH_STAMP(6);
   // Completion of the solve, part 1 (rmh_set_mass_completion): one Jacobi step x += D^-1 r with the residual the PCG
   // recurrence leaves behind -- no mass apply, no reduction; on the nearly affine elements of a refined mesh, where
   // D^-1 M = I + O(h), it gains the solve about one more order in h (nothing for a converged solve).
   if (HAS_HO && L.jacobi_step)
   {
#pragma unroll
      for (int r = 0; r < DR; r++) { xg[r] += rg[r] * dg[r]; }
   }
   // ---- phase J: back to Bernstein coefficients x_b = Ci (x) Ci (x) Ci x_g, stores --------------------------
   // (no barrier before the sA slot is overwritten: the PCG loop is left by all threads at the same point -- after
   // the barrier that follows its sA write, or after the barrier of its last reduction -- and nothing reads sA
   // between there and here)
#pragma unroll
   for (int r = 0; r < DR; r++)
   {
      const int t = tid + r * NT;
      if (t < NB * D3) { RMH_W(t / D3)[oSA + t % D3] = xg[r]; }
   }
   if (FUSED)
   {
      // limiter: the stencil extrema go to LDS behind the PCG buffers and the box table (smin at [PCG + 54, +27),
      // smax at [PCG + 81, +27)); the barriers of the back-transform publish them and the box table made from them
      static_assert(C::SLIM || C::W - C::PCG >= 108, "no room for the stencil and the box table behind the PCG buffers");
#pragma unroll
      for (int j = 0; j < NLS; j++)
      {
         const int k = tid + j * NT;
         if (k < NB * 27)
         {
            RMH_W(k / 27)[C::oLim + 54 + k % 27] = slo[j];
            RMH_W(k / 27)[C::oLim + 81 + k % 27] = shi[j];
         }
      }
   }
   __syncthreads();
   for (int dir = 0; dir < 3; dir++)
   {
      const int oin = (dir & 1) ? oSB : oSA, oout = (dir & 1) ? oSA : oSB;
      basis_leg(IntC<oCi>{}, dir, dir == 0, oin, oout);
      if (FUSED && dir == 0)
      {
      // per-dof bounds are box minima / maxima of the 3 x 3 x 3 stencil: a dof sees, per direction, the offsets
      // {-1, 0} on the low layer, {0} inside, {0, +1} on the high layer (remhos_tools.cpp:432-495).  The 27 boxes are
      // reduced once per element; every dof then reads one pair.  Bounds type 1: the same 7-point value for all.
      for (int k = tid; k < NB * 27; k += NT)
      {
         const int eb = k / 27, s3 = k % 27;
         const double *smin = RMH_W(eb) + C::oLim + 54, *smax = RMH_W(eb) + C::oLim + 81;
         double lo = INFINITY, hi = -INFINITY;
         if (L.bounds_type == 0)
         {
            // box {lo, hi} per direction: low layer {-1, 0} -> entries (0, 1), inside {0} -> (1, 1), high layer
            // {0, +1} -> (1, 2); the eight corners of the box are read unconditionally (duplicates are harmless).  Their
            // byte offsets come packed from the table (TabLayoutQ::oBoxQ) instead of from ~45 integer instructions
            if constexpr (C::ITAB)
            {
               const unsigned long long pk = (unsigned long long)__double_as_longlong(stab[C::oBoxQ + s3]);
#pragma unroll
               for (int c = 0; c < 8; c++)
               {
                  const int qb = (int)((pk >> (8 * c)) & 0xffu);
                  lo = fmin(lo, *(const double *)((const char *)smin + qb));
                  hi = fmax(hi, *(const double *)((const char *)smax + qb));
               }
            }
            else
            {
               const int sx = s3 % 3, sy = (s3 / 3) % 3, sz = s3 / 9;
               const int x0 = (sx == 0) ? 0 : 1, x1 = (sx == 2) ? 2 : 1;
               const int y0 = (sy == 0) ? 0 : 1, y1 = (sy == 2) ? 2 : 1;
               const int z0 = (sz == 0) ? 0 : 1, z1 = (sz == 2) ? 2 : 1;
#pragma unroll
               for (int c = 0; c < 8; c++)
               {
                  const int q = ((c & 1) ? x1 : x0) + 3 * ((c & 2) ? y1 : y0) + 9 * ((c & 4) ? z1 : z0);
                  lo = fmin(lo, smin[q]);
                  hi = fmax(hi, smax[q]);
               }
            }
         }
         else
         {
            constexpr int fs[7] = {13, 12, 14, 10, 16, 4, 22};
#pragma unroll
            for (int q = 0; q < 7; q++)
            {
               lo = fmin(lo, smin[fs[q]]);
               hi = fmax(hi, smax[fs[q]]);
            }
         }
         RMH_W(eb)[C::oLim + s3] = lo;
         RMH_W(eb)[C::oLim + 27 + s3] = hi;
      }
      }
      // (p = 3: the directions hand over within the wavefront that owns the element; one workgroup barrier at the end
      // publishes the box table)
      __syncthreads();
      if (dir == 2)
      {
         // (the last direction's outputs are in LDS, in the plain layout: the dof threads pick up theirs)
#pragma unroll
         for (int r = 0; r < DR; r++)
         {
            const int t = tid + r * NT;
            if (t < NB * D3) { xg[r] = RMH_W(t / D3)[oout + t % D3]; }
         }
      }
   }
   RMH_STAMP(16);
#pragma unroll
   for (int r = 0; r < DR; r++) { itmax = max(itmax, its[r]); }
   // Completion of the solve, part 2: the constant mode.  The integral of du_HO over the element is sum_i m_i x_i
   // (m: lumped mass, x: Bernstein coefficients), that of the exact solution 1^T b; the constant (1^T b - sum m x) /
   // |element| closes the gap, so that every stage conserves the mass to round-off for ANY stopping rule of the PCG --
   // and whatever the round-off of the back-transform was.  Both kept sums come from the PCG prelude.
   if (!FUSED)
   {
      if (L.mass_fix)
      {
#pragma unroll
         for (int r = 0; r < DR; r++) { tmp[r] = mm[r] * xg[r]; }
         batch_dot<C>(tid, tmp, red, lds, s_acc, ring);
         const double fix0_uni = UNI ? fdiv(RMH_W(0)[C::oKeep] - red[0], RMH_W(0)[C::oKeep + 1]) : 0.0; // (see UNI in the PCG loop)
#pragma unroll
         for (int r = 0; r < DR; r++)
         {
            const int t = tid + r * NT;
            if (t < NB * D3) { xg[r] += UNI ? fix0_uni : fdiv(RMH_W(t / D3)[C::oKeep] - red[r], RMH_W(t / D3)[C::oKeep + 1]); }
         }
      }
#pragma unroll
      for (int r = 0; r < DR; r++)
      {
         const int t = tid + r * NT;
         if (t < NB * D3 && e0 + t / D3 < L.e_end)
         {
            L.du[(size_t)e0 * D3 + t] = xg[r];
            L.m[(size_t)e0 * D3 + t] = mm[r];
         }
      }
      if (!C::WAVE_ALIGNED && NB == 1)
      {
         if (tid == 0 && e0 < L.e_end)
         {
            const double *part = s_acc + 4 * NB + 2; // (written in phase B, many barriers ago)
            double lo = part[0], hi = part[1];
#pragma unroll
            for (int w = 1; w < NT / 64; w++) { lo = fmin(lo, part[2 * w]); hi = fmax(hi, part[2 * w + 1]); }
            L.xe_min[e0] = lo;
            L.xe_max[e0] = hi;
         }
      }
      else if (!C::WAVE_ALIGNED && tid < NB && e0 + tid < L.e_end)
      {
         L.xe_min[e0 + tid] = my_min;
         L.xe_max[e0 + tid] = my_max;
      }
   }
   else
   {
      // ---- phase K: LimitMult + RK update (W is free: the last back-transform leg ended with a barrier) ----
      constexpr double eps = 1.0e-15;
      RMH_STAMP(21);
      // MassBasedAvg: ubar = sum m (u + dt du_HO) / sum m
      // (with the constant-mode completion: du_HO += c, and the element's new mass is sum m u + dt 1^T b)
      double mass[DR], vol[DR];
#pragma unroll
      for (int r = 0; r < DR; r++)
      {
         tmp[r] = mm[r] * uu[r];
         red[r] = mm[r] * xg[r];
      }
      if ((tid & 63) == 0 && itmax > cg_known) { atomicMax(L.cg_iters, itmax); }
      batch_dot2<C>(tid, tmp, red, mass, vol, lds, s_acc, ring);
      double fix_uni = 0.0;
#pragma unroll
      for (int r = 0; r < DR; r++)
      {
         const int t = tid + r * NT;
         const double sb = (t < NB * D3) ? RMH_W(t / D3)[C::oKeep] : 0.0, ev = (t < NB * D3) ? RMH_W(t / D3)[C::oKeep + 1] : 1.0;
         const double rate = L.mass_fix ? sb : vol[r]; // integral of du_HO over the element
         if (UNI) { if (r == 0) { fix_uni = fdiv(sb - vol[0], ev); } }
         xg[r] += L.mass_fix ? ((UNI && t < NB * D3) ? fix_uni : fdiv(sb - vol[r], ev)) : 0.0;
         mass[r] += L.dt * rate;
         vol[r] = ev;
      }
      RMH_STAMP(22);
      RMH_STAMP(17);
      double fcl[DR], pos[DR], neg[DR];
      double dtc = INFINITY; // UpdateTimeStepEstimate(u, du_LO, u_min, u_max), remhos.cpp:1839-1842
      const double ubar_uni = UNI ? fdiv(mass[0], vol[0]) : 0.0; // (element average: one division per element, see UNI in the PCG loop)
      const bool want_dt = L.dt_est != nullptr; // (uniform: without -dtc the candidates -- two IEEE divisions per dof -- are not formed)
      const double r_dt = fdiv_rcp(L.dt);
#pragma unroll
      for (int r = 0; r < DR; r++)
      {
         const int t = tid + r * NT;
         fcl[r] = 0.0; pos[r] = 0.0; neg[r] = 0.0;
         if (t < NB * D3)
         {
            const int eb = t / D3, i = t % D3;
            int s3; // class of the dof (low layer / inside / high layer per direction)
            if constexpr (C::ITAB) { s3 = ((const unsigned char *)(stab + C::oCls))[i]; }
            else
            {
               const int bx = i % D, by = (i / D) % D, bz = i / D2;
               s3 = (bx == 0 ? 0 : (bx == P ? 2 : 1)) + 3 * (by == 0 ? 0 : (by == P ? 2 : 1)) + 9 * (bz == 0 ? 0 : (bz == P ? 2 : 1));
            }
            const double lo = RMH_W(eb)[C::oLim + s3], hi = RMH_W(eb)[C::oLim + 27 + s3];
            const double ubar = UNI ? ubar_uni : fdiv(mass[r], vol[r]);
            if (!BOTH) { dlo[r] = fdiv_by(ubar - uu[r], L.dt, r_dt); } // MassBasedAvg; with RD dlo is already there
            if (want_dt) { dtc = fmin(dtc, dt_candidate(uu[r], dlo[r], lo, hi)); }
            const double u_new_lo = uu[r] + L.dt * dlo[r];
            const double m_dt = fdiv_by(mm[r], L.dt, r_dt);
            const double f_clip_min = m_dt * (lo - u_new_lo);
            const double f_clip_max = m_dt * (hi - u_new_lo);
            double fc = mm[r] * (xg[r] - dlo[r]);
            fc = fmin(f_clip_max, fmax(f_clip_min, fc));
            fcl[r] = fc;
            neg[r] = fmin(fc, 0.0);
            pos[r] = fmax(fc, 0.0);
         }
      }
      if (want_dt)
      {
         dtc = wave_minmax<true>(dtc);
         if ((tid & 63) == 63) { atomic_min_nonneg(L.dt_est, dtc); }
      }
      RMH_STAMP(18);
      double sumPos[DR], sumNeg[DR];
      batch_dot2<C>(tid, pos, neg, sumPos, sumNeg, lds, s_acc, ring);
      RMH_STAMP(19);
      double ynew[DR];
      const double rden_uni = UNI ? fdiv_rcp((sumNeg[0] + sumPos[0] > eps) ? sumPos[0] : sumNeg[0]) : 0.0;
      // (late arguments of the rounds below, read once: see the PCG prelude)
      const bool has_xb = L.x_base != nullptr;
      const double rk_aL = L.rk_a, rk_bL = L.rk_b, dt_rkL = L.dt_rk;
      const int e_endL = L.e_end;
      double *const y_outL = L.y_out, *const duL = L.du;
#pragma unroll
      for (int r = 0; r < DR; r++)
      {
         const int t = tid + r * NT;
         ynew[r] = 0.0;
         if (t < NB * D3)
         {
            const double new_mass = sumNeg[r] + sumPos[r];
            double fc = fcl[r];
            {
               // the two rescale branches (remhos_fct.cpp:523-532) exclude each other: one division, operands selected first
               const bool up = new_mass > eps, dn = new_mass < -eps;
               const double fpos = fmax(0.0, fc), fneg = fmin(0.0, fc);
               const double den = up ? sumPos[r] : sumNeg[r];
               // (UNI: the denominator is an element sum -- its reciprocal is refined once per lane, fdiv_rcp / fdiv_by = fdiv bit for bit)
               const double q = UNI ? fdiv_by(up ? fpos * sumNeg[r] : fneg * sumPos[r], den, rden_uni) : fdiv(up ? fpos * sumNeg[r] : fneg * sumPos[r], den);
               if (up) { fc = fneg - q; }
               else if (dn) { fc = fpos - q; }
            }
            const double dui = dlo[r] + fdiv(fc, mm[r]);
            ynew[r] = (has_xb ? rk_aL * xb[r] : 0.0) + rk_bL * (uu[r] + dt_rkL * dui);
            if (e0 + t / D3 < e_endL)
            {
               y_outL[(size_t)e0 * D3 + t] = ynew[r];
               if (duL) { duL[(size_t)e0 * D3 + t] = dui; }
            }
         }
      }
      RMH_STAMP(20);
      // element extrema of the new state
      if (C::WAVE_ALIGNED && DR == 2)
      {
         // min and max of both rounds in one reduction: after the half-wave swap the lower / upper 32 lanes hold
         // round 0 / 1; max is carried as min of the negated values, the row swap then puts {min r0, -max r0,
         // min r1, -max r1} into the four 16-lane rows and the four DPP row steps reduce them together
         const bool has1 = tid + NT < NB * D3;
         double y0 = ynew[0], y1 = has1 ? ynew[DR == 2 ? 1 : 0] : INFINITY;
         double z0 = -ynew[0], z1 = has1 ? -ynew[DR == 2 ? 1 : 0] : INFINITY;
         swap32(y0, y1);
         swap32(z0, z1);
         double mn = fmin(y0, y1), nx = fmin(z0, z1);
         swap16(mn, nx);
         double x = fmin(mn, nx);
         x = dpp_minmax<0xB1, 0xF, true>(x);
         x = dpp_minmax<0x4E, 0xF, true>(x);
         x = dpp_minmax<0x141, 0xF, true>(x);
         x = dpp_minmax<0x140, 0xF, true>(x);
         const int lane = tid & 63, eb = (lane < 32) ? (tid >> 6) : NT / D3 + (tid >> 6);
         if ((lane & 15) == 15 && eb < NB && e0 + eb < L.e_end)
         {
            // (both base pointers as scalars, selected by value: a select between the two kernarg FIELDS is compiled
            // into a per-lane load of the pointer itself, with a full wait in front of the store)
            double *pmin = L.xe_min_out, *pmax = L.xe_max_out;
            double *dst = (lane & 16) ? pmax : pmin;
            dst[e0 + eb] = (lane & 16) ? -x : x;
         }
      }
      else if (C::WAVE_ALIGNED)
      {
#pragma unroll
         for (int r = 0; r < DR; r++)
         {
            const int t = tid + r * NT;
            const double lo = wave_minmax<true>(t < NB * D3 ? ynew[r] : INFINITY);
            const double hi = wave_minmax<false>(t < NB * D3 ? ynew[r] : -INFINITY);
            if ((tid & 63) == 63 && t < NB * D3 && e0 + t / D3 < L.e_end)
            {
               L.xe_min_out[e0 + t / D3] = lo;
               L.xe_max_out[e0 + t / D3] = hi;
            }
         }
      }
      else if (NB == 1)
      {
         // one element per workgroup (p = 4, 5, 6): every thread reduces its own rounds, every wavefront by DPP, the
         // wavefront results meet in the partial-sum slots -- one barrier and no round trip of the values through LDS
         // (min / max do not depend on the order)
         constexpr int NW = NT / 64;
         double lo = INFINITY, hi = -INFINITY;
#pragma unroll
         for (int r = 0; r < DR; r++)
         {
            const bool in = tid + r * NT < D3;
            lo = fmin(lo, in ? ynew[r] : INFINITY);
            hi = fmax(hi, in ? ynew[r] : -INFINITY);
         }
         if (RMH_PACKED_SUMS)
         {
            const double x = wave_min_negmax(lo, hi); // (min in lane 31, -max in lane 63)
            lo = x;
            hi = -x;
         }
         else
         {
            lo = wave_minmax<true>(lo);
            hi = wave_minmax<false>(hi);
         }
         if (NW == 1)
         {
            if (RMH_PACKED_SUMS)
            {
               if (tid == 31 && e0 < L.e_end) { L.xe_min_out[e0] = lo; }
               if (tid == 63 && e0 < L.e_end) { L.xe_max_out[e0] = hi; }
            }
            else if (tid == 63 && e0 < L.e_end)
            {
               L.xe_min_out[e0] = lo;
               L.xe_max_out[e0] = hi;
            }
         }
         else
         {
            // (the two ring slots the last element sums did NOT use: slower threads may still be reading those)
            double *slo_ = s_acc + 4 * NB + 8 + C::N2S + ring * NW, *shi_ = s_acc + 4 * NB + 8 + C::N2S + ((ring + 1) % 4) * NW;
            if (RMH_PACKED_SUMS)
            {
               if ((tid & 63) == 31) { slo_[tid >> 6] = lo; }
               if ((tid & 63) == 63) { shi_[tid >> 6] = hi; }
            }
            else if ((tid & 63) == 63) { slo_[tid >> 6] = lo; shi_[tid >> 6] = hi; }
            __syncthreads();
            if (tid == 0 && e0 < L.e_end)
            {
               double l2 = slo_[0], h2 = shi_[0];
#pragma unroll
               for (int w = 1; w < NW; w++) { l2 = fmin(l2, slo_[w]); h2 = fmax(h2, shi_[w]); }
               L.xe_min_out[e0] = l2;
               L.xe_max_out[e0] = h2;
            }
         }
      }
      else
      {
         __syncthreads();
#pragma unroll
         for (int r = 0; r < DR; r++)
         {
            const int t = tid + r * NT;
            if (t < NB * D3) { RMH_W(t / D3)[64 + t % D3] = ynew[r]; }
         }
         __syncthreads();
         if (D3 >= 64)
         {
            // one wavefront per 64-dof chunk, then the chunks of an element
            constexpr int CH = C::DOT_CH;
            double *part = s_acc + 4 * NB + 8 + C::N2S; // [NB][CH][2]
            const int lane = tid & 63, wave = tid >> 6;
            for (int k = wave; k < NB * CH; k += NT / 64)
            {
               const int i = (k % CH) * 64 + lane;
               const double x = RMH_W(k / CH)[64 + min(i, D3 - 1)];
               const double lo = wave_minmax<true>(x), hi = wave_minmax<false>(x);
               if (lane == 63) { part[2 * k] = lo; part[2 * k + 1] = hi; }
            }
            __syncthreads();
            if (tid < NB && e0 + tid < L.e_end)
            {
               double lo = INFINITY, hi = -INFINITY;
#pragma unroll
               for (int c = 0; c < CH; c++)
               {
                  lo = fmin(lo, part[2 * (tid * CH + c)]);
                  hi = fmax(hi, part[2 * (tid * CH + c) + 1]);
               }
               L.xe_min_out[e0 + tid] = lo;
               L.xe_max_out[e0 + tid] = hi;
            }
         }
         else if (tid < NB && e0 + tid < L.e_end)
         {
            double lo = INFINITY, hi = -INFINITY;
            for (int i = 0; i < D3; i++)
            {
               lo = fmin(lo, RMH_W(tid)[64 + i]);
               hi = fmax(hi, RMH_W(tid)[64 + i]);
            }
            L.xe_min_out[e0 + tid] = lo;
            L.xe_max_out[e0 + tid] = hi;
         }
      }
   }
   // diagnostics: max PCG iteration count over the launch.  A global atomic per wavefront on ONE
   // address serialises at the memory side (~5 ns each: 3 ms per launch at 500 k wavefronts), so the
   // atomic is issued only when it can raise the (monotone) maximum.
   // (fused stage: done at the start of the limiter phase, in front of the L2 warm-up loads)
   if (!FUSED && (tid0 & 63) == 0 && itmax > cg_known) { atomicMax(a.cg_iters, itmax); }
   RMH_STAMP(7);
   RMH_STAMP_FLUSH();
}

#undef RMH_W

} // namespace rmh
