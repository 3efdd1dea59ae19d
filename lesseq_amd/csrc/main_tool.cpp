// count / solve / classify executables: the reference's argv, stdout, stderr log and exit
// status (count/count.cpp:88-129, solve/solve.cpp:102-146, classify/classify.cpp:51-79).
// The tool is chosen by the program name.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/lesseq_hip.h"

int main(int argc, char **argv) {
	const char *base = strrchr(argv[0], '/');
	base = base ? base + 1 : argv[0];
	const char *tool = strstr(base, "solve") ? "solve" : (strstr(base, "classify") ? "classify" : "count");
	return lsq_cli_main(tool, argc, argv);
}
