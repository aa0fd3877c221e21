// Test program for include/kmodel.hpp: load a model directory, answer a file of k-mer strings through the
// vector<string> front door (kmodel.hpp:90) and the single-string overload (kmodel.hpp:100), print one answer per line.
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "kmodel.hpp"

int main(int argc, char **argv)
{
	if (argc < 3) return 2;
	KModel *km = load_model(argv[1]);
	std::ifstream in(argv[2]);
	std::vector<std::string> q;
	for (std::string line; std::getline(in, line);)
		if (!line.empty()) q.push_back(line);
	std::vector<int> occ = km->kmer_to_occ(q, 4);
	for (size_t i = 0; i < occ.size(); i++) std::cout << occ[i] << "\n";
	// the single-string overload must agree with the batch; so must a batch of mixed lengths (each string on its own)
	for (size_t i = 0; i < q.size() && i < 50; i++)
		if (km->kmer_to_occ(q[i]) != occ[i]) return 3;
	std::vector<std::string> mixed;
	for (size_t i = 0; i < q.size() && i < 200; i++) mixed.push_back(i % 3 == 1 ? q[i].substr(0, 20) : q[i]);
	std::vector<int> m2 = km->kmer_to_occ(mixed, 4);
	for (size_t i = 0; i < mixed.size(); i++)
		if (i % 3 != 1 && m2[i] != occ[i]) return 4;
	delete km;
	// written the way the reference's README and main.cpp are: std names unqualified (the reference header leaks
	// `using namespace std;`), Tools::get_file_name (main.cpp:146), a default-constructed KModel filled by load (kmodel.hpp:43, :209)
	string dir = argv[1];
	if (Tools::get_file_name("/a/b/" + Tools::get_file_name(dir)) != Tools::get_file_name(dir)) return 5;
	KModel fresh;
	fresh.load(dir);
	vector<string> few(q.begin(), q.begin() + (q.size() < 100 ? q.size() : 100));
	vector<int> again = fresh.kmer_to_occ(few);
	for (size_t i = 0; i < few.size(); i++)
		if (again[i] != occ[i]) return 6;
	fresh.load_model(dir);                                            // a second load replaces the first
	if (fresh.kmer_to_occ(few[0]) != occ[0]) return 7;
	cout.flush();
	return 0;
}
