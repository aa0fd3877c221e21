// examples/kmcex_main.cpp -- the reference's driver (main.cpp:64-150) against include/kmodel.hpp + libkmx.so.
//
//   kmcEx [options] <input_file_name> <output_file_name> <working_directory>
//     -k<len> -t<threads> -ci<min> -cs<max> -nh<hashes> -nb<arrays>         (main.cpp:46-51)
//
// Same flow: run the KMC counter on the FASTQ input (the reference shells out to ./kmc_api/kmc, main.cpp:137-140;
// here the binary is taken from $KMC_BIN or ./kmc_api/kmc and skipped when absent so that an existing KMC
// database at <output_file_name> is used), build the model on the GPU, print the summary, save it under
// <working_directory>/<basename(output_file_name)>.  Unlike the reference, main returns a proper status.
#include "kmodel.hpp"

#include <cstdio>
#include <cstring>
#include <sys/stat.h>
#include <unistd.h>

struct Params {
	int k = 31, num_hash = 7, num_bit = 5, ci = 1, cs = 1023, t = 4;
	std::string input, output, workdir = "/tmp";
};

static bool parse(int argc, char **argv, Params &p)
{
	if (argc < 4) return false;
	int i = 1;
	for (; i < argc && argv[i][0] == '-'; ++i) {
		const char *a = argv[i];
		if (!strncmp(a, "-nh", 3)) p.num_hash = atoi(a + 3);
		else if (!strncmp(a, "-nb", 3)) p.num_bit = atoi(a + 3);
		else if (!strncmp(a, "-ci", 3)) p.ci = atoi(a + 3);
		else if (!strncmp(a, "-cs", 3)) p.cs = atoi(a + 3);
		else if (!strncmp(a, "-t", 2)) p.t = atoi(a + 2);
		else if (!strncmp(a, "-k", 2)) p.k = atoi(a + 2);
	}
	if (argc - i < 3) return false;
	p.input = argv[argc - 3];
	p.output = argv[argc - 2];
	p.workdir = argv[argc - 1];
	return !p.input.empty() && !p.output.empty() && !p.workdir.empty();
}

int main(int argc, char **argv)
{
	Params p;
	if (!parse(argc, argv, p)) {
		std::cout << "kmcEx (MI355X): counted k-mer encoding & decoding\n"
		             "USAGE  kmcEx [options] <input_file_name|@list> <output_file_name> <working_directory>\n"
		             "       -k<len> (31) -t<threads> (4) -ci<min count> (1) -cs<max count> (1023) -nh<hashes> (7) -nb<arrays> (5)\n";
		return 2;
	}
	const char *env = getenv("KMC_BIN");
	std::string kmc = env ? env : "./kmc_api/kmc";
	if (access(kmc.c_str(), X_OK) == 0) {
		char cmd[4096];
		snprintf(cmd, sizeof cmd, "%s -k%d -t%d -ci%d -cs%d %s %s %s", kmc.c_str(), p.k, p.t, p.ci, p.cs, p.input.c_str(), p.output.c_str(), p.workdir.c_str());
		std::cout << cmd << std::endl;
		if (system(cmd) != 0) { std::cout << "the KMC counter failed" << std::endl; return 1; }
	} else {
		std::cout << "no KMC binary (" << kmc << "): using the existing database " << p.output << std::endl;
	}
	KModel *km = get_model(p.ci, p.cs, p.num_hash, p.num_bit);
	km->init(p.output);
	km->show_header_info();
	km->show_kmodel_info();
	const size_t slash = p.output.find_last_of('/');
	const std::string dir = p.workdir + "/" + (slash == std::string::npos ? p.output : p.output.substr(slash + 1));
	mkdir(dir.c_str(), 0777);
	km->save(dir);
	delete km;
	return 0;
}
