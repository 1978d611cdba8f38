# Convenience: the product library (cylindertag_amd/), the test kit (testkit/) and the CPU oracle (oracle/) in one go.
# `python -c 'import __graft_entry__ as g; g.build()'` does the same.
all:
	$(MAKE) -s -C cylindertag_amd -j4
	$(MAKE) -s -C testkit
	$(MAKE) -s -C oracle
clean:
	$(MAKE) -s -C cylindertag_amd clean
	$(MAKE) -s -C testkit clean
	$(MAKE) -s -C oracle clean
# the vector-instruction issue-cost microbenchmark (runs on the GPU box; not part of the product)
ubench:
	/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Wno-unused-value tools/ubench/valu_rate.hip -o tools/ubench/valu_rate
	/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/ubench/dep_latency.hip -o tools/ubench/dep_latency
	/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/ubench/read_bw.hip -o tools/ubench/read_bw
.PHONY: all clean ubench
