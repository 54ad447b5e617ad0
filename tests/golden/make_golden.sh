#!/bin/bash
# Regenerates the golden plans in this directory FROM THE REAL REFERENCE (authoring container only:
# needs /root/reference and oracle/_ref/ref_dump built by `make -C oracle ref`).
# Each *.plan holds: the GEMM-pair plan recorded by EffectiveHamiltonian::precompute(), the operator
# blocks it points to, psi, diag and sigma_ref = H psi computed by the reference's own
# TensorFunctions::operator() (Tasked BatchGEMMSeq replay).  *.log holds per-site / final energies.
set -e
cd "$(dirname "$0")"
export MKL_THREADING_LAYER=GNU
R=../../oracle/_ref/ref_dump
D=/root/reference/data
$R $D/N2.STO3G.FCIDUMP su2 200 10 ./n2su2 dump=0:4,2:5 iprint=0          # E = -107.654122447525 (test_dmrg_n2_sto3g.cpp:187)
$R $D/N2.STO3G.FCIDUMP sz 60 6 ./n2sz dump=1:6,2:4 iprint=0
$R $D/H10.STO6G.R1.8.FCIDUMP sz 50 6 ./h10szm50 dump=0:6,1:5,2:4 iprint=0
# Cr2/SVP structure-only plans (pair descriptors, no operator data), converted to .struct.npz by
#   python -c "from block2_preview_amd.planfile import *; write_struct_npz(out, read_plan(in))"
$R $D/CR2.SVP.FCIDUMP su2 250 2 ./cr2m250 struct=1:5,1:10,1:20,1:30,0:20 occ=$D/CR2.SVP.OCC noise=1e-5,1e-5 iprint=1
# effective-Hamiltonian level fixtures (operator infos, operator tensors incl. delayed ones, term list of H_eff,
# the reference's ConnectionInfo, psi, diag, sigma_ref): input AND expected output of the symbolic -> numeric layer
$R $D/N2.STO3G.FCIDUMP sz 40 4 ./e_n2sz eham=1:5 iprint=0
$R $D/N2.STO3G.FCIDUMP su2 60 4 ./e_n2su2 eham=0:3,1:5 iprint=0
$R $D/H10.STO6G.R1.8.FCIDUMP sz 30 4 ./e_h10sz eham=0:5,1:4 iprint=0
# perturbative-noise fixtures: the single-GEMM list EffectiveHamiltonian::perturbative_noise records
# (captured through a TensorFunctions subclass, oracle/ref_dump.cpp capture_pnoise), its operands and the perturbed
# wavefunctions the reference returned; sweep 0 runs forward (TraceTypes::Right), sweep 1 backward (TraceTypes::Left)
# (.enoise: the same step at the symbolic level — the eham content plus operator sub-labels, perturbed-ket infos, result)
$R $D/N2.STO3G.FCIDUMP su2 60 3 ./p_n2su2 pnoise=0:3,1:5 enoise=0:3,1:5 iprint=0
$R $D/H10.STO6G.R1.8.FCIDUMP sz 30 3 ./p_h10sz pnoise=0:5,1:4 enoise=1:4 iprint=0
# Cr2/SVP M=250 noise lists, structure only; converted to .pnoise_struct.npz by
#   python -c "from block2_preview_amd.planfile import *; write_gemm_struct_npz(out, read_gemm_list(in))"
$R $D/CR2.SVP.FCIDUMP su2 250 2 ./cr2m250 pnoise_struct=0:20,1:20,1:10 occ=$D/CR2.SVP.OCC noise=1e-5,1e-5 iprint=1
# environment-rotation fixtures: the GEMM pairs the reference's own tensor_rotate records (Auto mode)
# for TensorFunctions::left_rotate / right_rotate, the enlarged operators, the MPS tensor and the rotated operators the
# reference computed (.plan), plus the same step at the symbolic level (.erot: operator infos, MPS tensor infos)
# (rotation and blocking fixtures come from ONE run per molecule, see the blocking entry below, so that the enlarged
#  operators the blocking step produces are exactly the rotation's input: tests chain them on the device)
# Cr2/SVP M=250 rotation structures -> *.rotstruct.npz (write_struct_npz)
$R $D/CR2.SVP.FCIDUMP su2 250 2 ./rcr2 rot_struct=0:20,1:20 occ=$D/CR2.SVP.OCC noise=1e-5,1e-5 iprint=1
# blocking fixtures: the element-wise block-product terms re-grouped from the k = 1 GEMM groups the
# reference's own TensorFunctions::tensor_product records (Auto mode) for left_contract / right_contract, the block and
# site operators and the enlarged operators the reference computed
# (.eblk: the same step at the symbolic level — operator infos incl. tensor-product connection infos, expressions)
$R $D/N2.STO3G.FCIDUMP su2 60 3 ./x_n2su2 blk=0:4,1:4 eblk=0:4,1:4 rot=0:4,1:4 erot=0:4,1:4 iprint=0
$R $D/H10.STO6G.R1.8.FCIDUMP sz 30 3 ./x_h10sz blk=0:5 eblk=0:5 rot=0:5,1:4 erot=0:5 iprint=0
for f in x_*; do case $f in *blk*) n=blk_${f#x_};; *rot*) n=rot_${f#x_};; *) n=lcr_${f#x_};; esac; mv $f $n; done
# Cr2/SVP M=250 blocking structures -> *.blkstruct.npz (planfile.write_outer_struct_npz)
$R $D/CR2.SVP.FCIDUMP su2 250 2 ./bcr2 blk_struct=0:20,1:20 occ=$D/CR2.SVP.OCC noise=1e-5,1e-5 iprint=1
# on-disk format fixtures: MPS tensors written by the reference's own SparseMatrix::save_data(file, true)
# next to their content as named arrays (.arr)
$R $D/N2.STO3G.FCIDUMP su2 60 2 ./disk_n2su2 tensor_file=3,6 iprint=0
$R $D/H10.STO6G.R1.8.FCIDUMP sz 30 2 ./disk_h10sz tensor_file=4 iprint=0
# BASELINE configs[4]: 1D Hubbard L=16 U/t=4 (HubbardFCIDUMP, src/core/hubbard.hpp:30-82), SZ: golden plans with data at
# M=100 and the TRUE M=3000 structure of the two-site step 7-8 in sweep 0 (fixed-M random MPS; the run stops after the
# capture) -> hubbard_l16_u4_sz_m3000_sw0_site7.struct.npz
$R hubbard:16:1:4 sz 100 4 ./hubu4m100 dump=1:7,2:3 iprint=0 noise=1e-5,1e-5,1e-6,0
$R hubbard:16:1:4 sz 3000 1 ./hub3000 struct=0:7 stop_after=0:7 dav_iter=2 iprint=2
# BASELINE configs[1] at its true size: H10/STO-6G SZ M=500 mid-chain structure -> h10_sz_m500_sw1_site4.struct.npz
$R $D/H10.STO6G.R1.8.FCIDUMP sz 500 2 ./h10m500 struct=1:4 stop_after=1:4 iprint=1
# sum-MPO partition: the reference on 2 MPI ranks with ParallelRuleSimple(IJ) (make -C oracle _ref/ref_dump_mpi):
# every rank's own plan + operators, psi, and sigma_ref = the ALL-REDUCED H psi; *.prefactors = index_prefactor tables
/opt/conda/bin/mpirun -n 2 ../../oracle/_ref/ref_dump_mpi $D/N2.STO3G.FCIDUMP su2 200 6 ./n2su2_ij para=ij prefactors=1 dump=1:5,2:4 nthreads=4 iprint=0
# the site-to-site chain (tests/test_sweep_gpu.py): every blocking / rotation / operator-sum / effective-Hamiltonian event
# of the initial environments and of sweeps 0 and 1 of N2/STO-3G SU2 M=200 at the symbolic level, in call order, without
# operator or wavefunction data (delayed contraction and the contraction cache off: every enlarged block is contracted in
# full; no noise, tight Davidson, so that the site energies are a deterministic function of the chain)
mkdir -p chain_n2su2
$R $D/N2.STO3G.FCIDUMP su2 200 2 ./chain_n2su2/n2c chain=1 nodelay=1 nocache=1 noise=0,0 tol=1e-12 iprint=0
# the same for an SZ system (BASELINE configs[1]'s molecule AT ITS BASELINE BOND DIMENSION M=500; the run converges in 2 sweeps): 114 events, 18 site energies
mkdir -p chain_h10sz
$R $D/H10.STO6G.R1.8.FCIDUMP sz 500 4 ./chain_h10sz/h10c chain=3 nodelay=1 nocache=1 noise=0,0,0,0 tol=1e-12 iprint=0
# the 2-rank sum-MPO run (ParallelRuleSimple IJ) with the event chain of EVERY rank: n2p.r{0,1}of2.*
mkdir -p chain_n2su2_ij
/opt/conda/bin/mpirun -n 2 ../../oracle/_ref/ref_dump_mpi $D/N2.STO3G.FCIDUMP su2 200 2 ./chain_n2su2_ij/n2p para=ij chain=1 nocache=1 nthreads=2 noise=0,0 tol=1e-12 iprint=0
# ... and a 4-rank run
mkdir -p chain_n2su2_ij4
/opt/conda/bin/mpirun -n 4 ../../oracle/_ref/ref_dump_mpi $D/N2.STO3G.FCIDUMP su2 200 2 ./chain_n2su2_ij4/n2p para=ij chain=1 nocache=1 nthreads=1 noise=0,0 tol=1e-12 iprint=0
# ... and with the noisy schedule (every rank records its own perturbative-noise step; the perturbed labels are the union over
# the ranks); kept as one archive per rank
mkdir -p chain_n2su2_ij_noisy
/opt/conda/bin/mpirun -n 2 ../../oracle/_ref/ref_dump_mpi $D/N2.STO3G.FCIDUMP su2 200 3 ./chain_n2su2_ij_noisy/n2pn para=ij chain=2 nocache=1 nthreads=2 noise=1e-5,1e-5,0 tol=1e-12 dav_thrd=1e-13 iprint=0
(cd chain_n2su2_ij_noisy && for r in 0 1; do zip -q -9 n2pn.r${r}of2.zip n2pn.r${r}of2.* && rm -f n2pn.r${r}of2.ev* n2pn.r${r}of2.log; done)
# Cr2/SVP at M=30, two sweeps: 539 events, 51 MB raw -> kept as chain_cr2/cr2c.zip (sweep.ChainFixture unpacks it)
# (round 3: spectra=1 logs the full density-matrix spectrum the reference truncated at every bond, dav_thrd the Davidson
#  threshold of every sweep; SWEEP_TIME lines = the reference's own per-sweep timers)
zipchain() { (cd $1 && python3 -c "import zipfile,glob,os; p='$2'; z=zipfile.ZipFile(p+'.zip','w',zipfile.ZIP_DEFLATED,compresslevel=9); [z.write(f) for f in sorted(glob.glob(p+'.ev*')+[p+'.log'])]; z.close()"); }
mkdir -p chain_cr2 /tmp/b2x_cr2c
$R $D/CR2.SVP.FCIDUMP su2 30 2 /tmp/b2x_cr2c/cr2c chain=1 nodelay=1 nocache=1 noise=0,0 tol=1e-12 dav_thrd=1e-13 spectra=1 iprint=0 occ=$D/CR2.SVP.OCC
zipchain /tmp/b2x_cr2c cr2c && cp /tmp/b2x_cr2c/cr2c.zip chain_cr2/
# THE Cr2 gate: M=250, the reference's noisy schedule, three sweeps, with a cut-off of 1e-9 on the kept
# density-matrix weights (DMRG::cutoff) and Davidson converged to 1e-18: 845 events -> chain_cr2_m250_cut9/cr2g.zip (74 s on 8 threads)
mkdir -p chain_cr2_m250_cut9 /tmp/b2x_cr2g
$R $D/CR2.SVP.FCIDUMP su2 250 3 /tmp/b2x_cr2g/cr2g chain=2 nodelay=1 nocache=1 noise=1e-5,1e-5,0 tol=1e-12 dav_thrd=1e-18 cutoff=1e-9 spectra=1 iprint=0 occ=$D/CR2.SVP.OCC nthreads=8
zipchain /tmp/b2x_cr2g cr2g && cp /tmp/b2x_cr2g/cr2g.zip chain_cr2_m250_cut9/
# ... and at M=500 (two sweeps: one noisy, one noise-free; 582 events, 154 s on 8 threads) -> chain_cr2_m500_cut9/cr2h.zip
mkdir -p chain_cr2_m500_cut9 /tmp/b2x_cr2h
$R $D/CR2.SVP.FCIDUMP su2 500 2 /tmp/b2x_cr2h/cr2h chain=1 nodelay=1 nocache=1 noise=1e-5,0 tol=1e-12 dav_thrd=1e-18 cutoff=1e-9 spectra=1 iprint=0 occ=$D/CR2.SVP.OCC nthreads=8
zipchain /tmp/b2x_cr2h cr2h && cp /tmp/b2x_cr2h/cr2h.zip chain_cr2_m500_cut9/
# The same WITHOUT the cut-off (block2's default 1e-14) and dav_thrd=1e-13 is ill-posed — kept states reach into the numerical null
# space of the density matrix: the inputs of profiles/r03_cr2_m250_noisy_trunc_diag.txt (noise=1e-5,1e-5,0; 693 s on 3 threads),
# r03_cr2_m250_noise_free_trunc_diag.txt (noise=0,0) and r03_reference_reproducibility_cr2_m250.txt (nthreads=8 / 5 / 3, no chain=);
# those chains are not committed (2 x 28 MB).
# $R $D/CR2.SVP.FCIDUMP su2 250 3 /tmp/b2x_cr2n250/cr2n250 chain=2 nodelay=1 nocache=1 noise=1e-5,1e-5,0 tol=1e-12 dav_thrd=1e-13 spectra=1 iprint=1 occ=$D/CR2.SVP.OCC nthreads=3
# noisy schedules on N2 and H10 (noises 1e-5, 1e-5, 0: perturbative noise -> perturbed density matrix -> split in the chain)
mkdir -p chain_n2su2_noisy chain_h10sz_noisy
$R $D/N2.STO3G.FCIDUMP su2 200 3 ./chain_n2su2_noisy/n2n chain=2 nodelay=1 nocache=1 noise=1e-5,1e-5,0 tol=1e-12 dav_thrd=1e-13 iprint=0 spectra=1 nthreads=2
$R $D/H10.STO6G.R1.8.FCIDUMP sz 500 3 ./chain_h10sz_noisy/h10n chain=2 nodelay=1 nocache=1 noise=1e-5,1e-5,0 tol=1e-12 dav_thrd=1e-13 iprint=0 spectra=1 nthreads=2
zipchain chain_n2su2_noisy n2n && rm -f chain_n2su2_noisy/n2n.ev* chain_n2su2_noisy/n2n.log
zipchain chain_h10sz_noisy h10n && rm -f chain_h10sz_noisy/h10n.ev* chain_h10sz_noisy/h10n.log
# TRUE Cr2/SVP structures above M=250 (fixed-M run from a random MPS, two Davidson iterations per site, captured at sweep 1
# site 20 like the M=250 plan; M=2000 needs stack_gb=40 and ~25 min on 5 threads) -> cr2_su2_m{1000,2000}_sw1_site20.struct.npz
$R $D/CR2.SVP.FCIDUMP su2 1000 2 /tmp/cr2m1000b struct=1:20 stop_after=1:20 occ=$D/CR2.SVP.OCC dav_iter=2 nthreads=4 iprint=2 noise=1e-5,1e-5
$R $D/CR2.SVP.FCIDUMP su2 2000 2 /tmp/cr2m2000 struct=1:20 stop_after=1:20 occ=$D/CR2.SVP.OCC dav_iter=2 nthreads=5 iprint=2 noise=1e-5,1e-5 stack_gb=40
# the bundled 1D Hubbard L=16 file at M=500: the run converges in 4 sweeps (325 events, 60 site energies; 13 MB)
mkdir -p chain_hubu2
$R $D/HUBBARD-L16.FCIDUMP sz 500 6 ./chain_hubu2/hubc chain=5 nodelay=1 nocache=1 noise=0,0,0,0,0,0 tol=1e-12 iprint=0
# compressed storage: the same MPS tensor written plain and through the reference's FPCodec (fp_prec, fp_chunk), and the
# names of the scratch files of a run that is left in the middle of sweep 1 (ls of the scratch directory)
$R $D/N2.STO3G.FCIDUMP su2 60 2 ./diskc_n2su2 tensor_file=4 fp_prec=1e-8 fp_chunk=64 iprint=0
$R $D/H10.STO6G.R1.8.FCIDUMP sz 30 2 ./diskc_h10sz tensor_file=5 fp_prec=1e-5 fp_chunk=1024 iprint=0
$R $D/N2.STO3G.FCIDUMP su2 60 2 ./x stop_after=1:4 iprint=0 scratch=/tmp/b2x_listing && ls /tmp/b2x_listing > scratch_listing.txt
# the Cr2 site-20 plan WITHOUT occupation-guided initial bond dimensions (what SURVEY.md counted: 400-570 k pairs) ->
# cr2_su2_m250_noocc_sw1_site20.struct.npz
$R $D/CR2.SVP.FCIDUMP su2 250 2 ./cr2noocc struct=1:20 stop_after=1:20 noise=1e-5,1e-5 iprint=0
# partition-file content: the files the reference writes while building the initial environments with its default
# stack allocation (main_stack=1); the run is left at the first site, so its scratch directory still holds them -> part_n2su2/
mkdir -p part_n2su2 /tmp/b2x_partchain
$R $D/N2.STO3G.FCIDUMP su2 200 2 /tmp/b2x_partchain/n2p chain=0 nodelay=1 nocache=1 noise=0,0 tol=1e-12 dav_thrd=1e-13 iprint=0 main_stack=1 stop_after=0:0 scratch=/tmp/b2x_part nthreads=2
cp /tmp/b2x_part/F0.PART.DMRG.RIGHT.* part_n2su2/ && zipchain /tmp/b2x_partchain n2p && cp /tmp/b2x_partchain/n2p.zip part_n2su2/
