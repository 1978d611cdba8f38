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
.PHONY: all clean
