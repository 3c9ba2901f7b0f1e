# Builds librepsurf_hip.so (gfx950 only) and the CPU oracle.  No cmake, no torch dependency.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := repsurf_amd/csrc
LIBDIR := repsurf_amd/lib
# -ffp-contract=off: the index-producing kernels restate the reference CPU arithmetic
# operation by operation; fused multiply-adds appear only where written (rs_fma).
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off \
            -fhip-fp32-correctly-rounded-divide-sqrt -munsafe-fp-atomics -Wall -Wno-unused-function
SRCS := $(CSRC)/rs_lib.cpp $(CSRC)/fps.hip $(CSRC)/ballquery.hip $(CSRC)/knn_umbrella.hip $(CSRC)/knn_wide.hip \
        $(CSRC)/group.hip $(CSRC)/interp.hip $(CSRC)/seg_geom.hip $(CSRC)/scene_knn.hip $(CSRC)/grid_knn.hip $(CSRC)/mlp.hip $(CSRC)/umbrella_mlp.hip $(CSRC)/umbrella_mfma.hip $(CSRC)/head.hip $(CSRC)/adam.hip
OBJS := $(patsubst $(CSRC)/%,build/%.o,$(SRCS)) build/mlp_bf16.hip.o build/mlp_sb.hip.o build/mlp_split.hip.o

# The geometry kernels run on a side stream BESIDE the network's MFMA kernels (graph.PipelinedStep / RaggedSegStep).  Packed-fp32 VALU
# instructions as the SLP vectorizer makes them of neighbouring float operations -- precisely: v_pk_add_f32 / v_pk_fma_f32 with op_sel on
# their second source -- gave other results in lanes 48..63 of a wave in ~1-2.5 % of a kernel's launches beside the split-product (bf16
# MFMA) GEMMs; 0 of 20 000 launches of the same kernel built without them, 0 alone on a stream (tools/victim_probe.py,
# profiles/r06/eager_beside_graph.txt).  Those translation
# units -- and, at no measured cost (tools/slp_ab.sh), every other one without MFMA loops -- are compiled without the SLP and loop vectorizers; their
# arithmetic is the same IEEE operations either way.
GEOM_TUS := fps ballquery knn_umbrella knn_wide group interp seg_geom scene_knn grid_knn umbrella_mlp head adam
$(foreach t,$(GEOM_TUS),$(eval build/$(t).hip.o: HIPFLAGS += -fno-slp-vectorize -fno-vectorize))

all: $(LIBDIR)/librepsurf_hip.so oracle

build/%.o: $(CSRC)/% $(CSRC)/rs_common.h $(CSRC)/umbrella_fan.h include/repsurf_hip.h
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

# mlp.hip is four translation units (fp32 entry points / the two bf16 ones / their bf16-storage instances / the split-product ones): they compile
# side by side
build/mlp.hip.o: HIPFLAGS += -DRS_MLP_TU=0
build/mlp_bf16.hip.o: $(CSRC)/mlp.hip $(CSRC)/rs_common.h include/repsurf_hip.h
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -DRS_MLP_TU=1 -x hip -c $< -o $@
# (the only GEMM unit where the vectorizer produced the instruction form of the note above -- 10 of them, in two weight-gradient instances; built
#  without it the bf16 lines read the same: tools/sb_ab.sh)
build/mlp_sb.hip.o: $(CSRC)/mlp.hip $(CSRC)/rs_common.h include/repsurf_hip.h
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -fno-slp-vectorize -DRS_MLP_TU=3 -x hip -c $< -o $@
# unit 4: the fp32 product as six bf16 MFMAs over three-part operands (the default of the tiled kernels; RS_GEMM_SPLIT3=0: fp32 MFMAs)
build/mlp_split.hip.o: $(CSRC)/mlp.hip $(CSRC)/rs_common.h include/repsurf_hip.h
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -DRS_MLP_TU=4 -x hip -c $< -o $@

$(LIBDIR)/librepsurf_hip.so: $(OBJS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

oracle: oracle/_build/libgeom_oracle.so
oracle/_build/libgeom_oracle.so: oracle/geom_oracle.c
	@mkdir -p oracle/_build
	gcc -O2 -std=c11 -fPIC -shared -ffp-contract=off -fno-fast-math -fopenmp -o $@ $< -lm

# the reference's own pointops kernels as host code (test infrastructure; needs /root/reference)
ref:
	$(MAKE) -f oracle/Makefile.ref

clean:
	rm -rf build $(LIBDIR)/librepsurf_hip.so oracle/_build

.PHONY: all oracle ref clean
