# Build everything that is native in this repo.
#   make lib      -> ffmpeg-ht_amd/libhtj2k_amd.so   (product: host parser + HIP kernels, gfx950)
#   make oracle   -> oracle/libj2k_oracle.so          (tests only: CPU restatement of the reference)
#   make vecgen   -> tools/vecgen/libhtj2k_vecgen.so  (tests only: HTJ2K test-vector factory)
HIPCC   ?= /opt/rocm/bin/hipcc
CC      ?= gcc
ARCH    ?= gfx950
CFLAGS  ?= -O2 -g -Wall -Wextra -fPIC
PKG     := ffmpeg-ht_amd
CSRC    := $(PKG)/csrc

all: lib oracle vecgen

lib: $(PKG)/libhtj2k_amd.so
oracle: oracle/libj2k_oracle.so
vecgen: tools/vecgen/libhtj2k_vecgen.so
ubench: tools/ubench/membw tools/ubench/occupancy tools/ubench/valu_rate
examples: examples/htj2k_decode

examples/htj2k_decode: examples/htj2k_decode.c include/htj2k_amd.h $(PKG)/libhtj2k_amd.so
	$(CC) -O2 -Wall -std=gnu11 -Iinclude -o $@ $< -L$(PKG) -lhtj2k_amd -Wl,-rpath,'$$ORIGIN/../$(PKG)'

tools/ubench/membw: tools/ubench/membw.hip
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -o $@ $<

tools/ubench/valu_rate: tools/ubench/valu_rate.hip
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -o $@ $<

tools/ubench/occupancy: tools/ubench/occupancy.hip $(wildcard $(CSRC)/*.hpp)
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -o $@ $<

# host front-end: codestream syntax, geometry + Tier-2, plan assembly
HOSTOBJ := $(CSRC)/j2k_syntax.o $(CSRC)/j2k_tier2.o $(CSRC)/j2k_plan.o
$(HOSTOBJ): %.o: %.c $(CSRC)/j2k_host.h $(CSRC)/j2k_plan.h include/htj2k_amd.h
	$(CC) $(CFLAGS) -std=gnu11 -c $< -o $@

# -ffp-contract=off: the reference objects contain no FMA; 9/7 parity needs
# separately rounded multiply and add.
$(CSRC)/htj2k_device.o: $(CSRC)/htj2k_device.hip $(CSRC)/j2k_plan.h $(CSRC)/ht_cxtvlc_rows.h include/htj2k_amd.h $(wildcard $(CSRC)/*.hpp)
	$(HIPCC) --offload-arch=$(ARCH) -O3 -g -fPIC -ffp-contract=off -std=c++17 -Wall $(HIPFLAGS_EXTRA) -c $< -o $@

$(CSRC)/htj2k_pipe.o: $(CSRC)/htj2k_pipe.cpp include/htj2k_amd.h
	$(CXX) -O2 -g -fPIC -std=c++17 -Wall -pthread -c $< -o $@

$(CSRC)/j2k_split.o: $(CSRC)/j2k_split.c include/htj2k_amd.h
	$(CC) $(CFLAGS) -std=gnu11 -c $< -o $@

$(CSRC)/j2k_mxf.o: $(CSRC)/j2k_mxf.c include/htj2k_amd.h
	$(CC) $(CFLAGS) -std=gnu11 -c $< -o $@

$(PKG)/libhtj2k_amd.so: $(CSRC)/htj2k_device.o $(CSRC)/htj2k_pipe.o $(HOSTOBJ) $(CSRC)/j2k_split.o $(CSRC)/j2k_mxf.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $^ -lpthread

# The oracle is self-contained: its own parser (oracle/j2k_oracle_parse.c, a close restatement of the
# reference's) and its own CPU restatement of the reference's compute.  Nothing of csrc/ is compiled in
# except the constant CxtVLC rows.
oracle/libj2k_oracle.so: oracle/j2k_oracle.c oracle/j2k_oracle_mq.c oracle/j2k_oracle_parse.c oracle/j2k_oracle_split.c oracle/j2k_oracle_plan.h $(CSRC)/ht_cxtvlc_rows.h
	$(CC) -O3 -fno-math-errno -fno-signed-zeros -fno-tree-vectorize -ffp-contract=off -g -Wall -Wextra -fPIC -std=gnu11 \
	    -shared -o $@ oracle/j2k_oracle.c oracle/j2k_oracle_mq.c oracle/j2k_oracle_parse.c oracle/j2k_oracle_split.c -lm

tools/vecgen/libhtj2k_vecgen.so: tools/vecgen/htj2k_enc.c tools/vecgen/htj2k_enc.h $(CSRC)/ht_cxtvlc_rows.h
	$(CC) $(CFLAGS) -std=gnu11 -shared -o $@ $< -lm

clean:
	rm -f $(CSRC)/*.o $(PKG)/*.so oracle/*.so tools/vecgen/*.so

.PHONY: all lib oracle vecgen ubench examples clean
