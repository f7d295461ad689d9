// Writes the INPUT tuples of the reference's randomized property tests
// (tests/test_multiply_sparse.cpp:84-95 MM, :138-149 MV; seeds 1..999, dsize 5)
// so the tests need no libstdc++ at run time.  Own code: only the recipe for
// drawing the inputs is restated (std::default_random_engine seeded with
// `seed`, std::bind COPIES the engine so the index and the value distribution
// each replay the stream from the seed).  The expected outputs are not stored:
// the property is "sparse product == dense triple loop", checked by the test.
//
// Output (text, one record per line):
//   MM <seed> A <n> (i j hexval)* B <n> (i j hexval)*
//   MV <seed> A <n> (i j hexval)* V <n> (i hexval)*
#include <cstdio>
#include <functional>
#include <random>
#include <vector>

int main()
{
	const unsigned dsize = 5;
	for (int seed = 1; seed < 1000; ++seed) {
		std::default_random_engine generator(seed);
		auto dim_distro(std::bind(std::uniform_int_distribution<int>(0, dsize - 1), generator));
		auto val_distro(std::bind(std::uniform_real_distribution<double>(0, 1), generator));
		std::printf("MM %d", seed);
		for (int m = 0; m < 2; ++m) {
			int n = (int)(val_distro() * (double)(dsize * dsize));
			std::printf(" %c %d", m == 0 ? 'A' : 'B', n);
			for (int t = 0; t < n; ++t) {
				int i = dim_distro();
				int j = dim_distro();
				double v = val_distro();
				std::printf(" %d %d %a", i, j, v);
			}
		}
		std::printf("\n");
	}
	for (int seed = 1; seed < 1000; ++seed) {
		std::default_random_engine generator(seed);
		auto dim_distro(std::bind(std::uniform_int_distribution<int>(0, dsize - 1), generator));
		auto val_distro(std::bind(std::uniform_real_distribution<double>(0, 1), generator));
		std::printf("MV %d", seed);
		int n = (int)(val_distro() * (double)(dsize * dsize));
		std::printf(" A %d", n);
		for (int t = 0; t < n; ++t) {
			int i = dim_distro();
			int j = dim_distro();
			double v = val_distro();
			std::printf(" %d %d %a", i, j, v);
		}
		n = (int)(val_distro() * (double)dsize);
		std::printf(" V %d", n);
		for (int t = 0; t < n; ++t) {
			int i = dim_distro();
			double v = val_distro();
			std::printf(" %d %a", i, v);
		}
		std::printf("\n");
	}
	return 0;
}
