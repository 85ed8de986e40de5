#!/usr/bin/env python3
"""lab: writes build/lab/linpsf_mfma_stamped.hip = csrc/linpsf_mfma.hip with s_memtime stamps around the phases of the matrix-core
LinPSF fit (sums over all wavefronts, printed by the launch function), and builds tools/lab/libtessphot_fitm_stamp.so."""
import os, subprocess, glob
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
csrc = os.path.join(root, 'photometry_amd', 'csrc')
s = open(os.path.join(csrc, 'linpsf_mfma.hip')).read()
def rep(old, new, count=1):
	global s
	assert old in s, old
	s = s.replace(old, new, count)
rep('typedef double f64x4 __attribute__((ext_vector_type(4)));\n', '''typedef double f64x4 __attribute__((ext_vector_type(4)));
__device__ unsigned long long g_fitm_stamps[8];
#define STAMP(i) do { const unsigned long long _t = __builtin_amdgcn_s_memtime(); if (lane == 0) lab_t[i] += _t - lab_last; lab_last = __builtin_amdgcn_s_memtime(); } while (0)
''')
rep('	const float c2f = (float)c2;\n', '	const float c2f = (float)c2;\n	unsigned long long lab_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};\n	unsigned long long lab_last = __builtin_amdgcn_s_memtime();\n')
rep('	if (ntiles > 0) stage_tile(0, bst[0]);\n', '	STAMP(0);\n	if (ntiles > 0) stage_tile(0, bst[0]);\n')
rep('		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n		__syncthreads();   // tile P has landed', '		STAMP(1);\n		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n		__syncthreads();   // tile P has landed')
rep('		const float* buf = bst[P & 1];\n', '		const float* buf = bst[P & 1];\n		STAMP(2);\n')
rep('			// normal equations of the tile:', '			STAMP(3);\n			// normal equations of the tile:')
rep("""					for (int t = s; t < S; ++t) { acc[c][m] += av[s] * av[t]; ++m; }
				}
			}
		}
	}
""", """					for (int t = s; t < S; ++t) { acc[c][m] += av[s] * av[t]; ++m; }
				}
			}
			STAMP(4);
		}
	}
""")
rep('	const int k = w0 + kl;\n	if (g >= NT', '	const int k = w0 + kl;\n	STAMP(6);\n	if (lane == 0) for (int i = 0; i < 8; ++i) atomicAdd(&g_fitm_stamps[i], lab_t[i]);\n	if (g >= NT')
rep('	TP_FITM(1);\n', '''	{
		unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t[3][8];
		(void)hipMemcpyToSymbol(HIP_SYMBOL(g_fitm_stamps), z, sizeof(z));
		TP_FITM(1); (void)hipStreamSynchronize(ctx->stream); (void)hipMemcpyFromSymbol(t[0], HIP_SYMBOL(g_fitm_stamps), sizeof(z)); (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fitm_stamps), z, sizeof(z));
		TP_FITM(2); (void)hipStreamSynchronize(ctx->stream); (void)hipMemcpyFromSymbol(t[1], HIP_SYMBOL(g_fitm_stamps), sizeof(z)); (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fitm_stamps), z, sizeof(z));
		TP_FITM(3); (void)hipStreamSynchronize(ctx->stream); (void)hipMemcpyFromSymbol(t[2], HIP_SYMBOL(g_fitm_stamps), sizeof(z));
		const char* nm[7] = {"setup", "K load issue", "wait + barrier", "tile reads + B + MFMA", "accumulate", "end barrier", "reduce + solve"};
		for (int i = 0; i < 7; ++i) fprintf(stderr, "STAMP %-24s S1 %8.2f  S2 %8.2f  S3 %8.2f  G wave-ticks\\n", nm[i], t[0][i] * 1e-9, t[1][i] * 1e-9, t[2][i] * 1e-9);
		return TP_OK;
	}
	TP_FITM(1);
''')
os.makedirs(os.path.join(csrc, 'build', 'lab'), exist_ok=True)
out = os.path.join(csrc, 'build', 'lab', 'linpsf_mfma_stamped.hip')
open(out, 'w').write(s)
obj = out[:-4] + '.o'
subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-ffp-contract=off', '-I' + os.path.join(root, 'include'), '-I' + csrc,
	'-x', 'hip', '-c', out, '-o', obj], check=True)
objs = [o for o in glob.glob(os.path.join(csrc, 'build', '*.o')) if 'linpsf_mfma' not in o]
subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', os.path.join(root, 'tools', 'lab', 'libtessphot_fitm_stamp.so')] + objs + [obj,
	'-L/opt/rocm/lib', '-lrccl'], check=True)
print('built tools/lab/libtessphot_fitm_stamp.so')
