#include "sd_cli.h"
namespace sdcli {
int searchModule(const Args &) { return fail("not built yet"); }
int clustersearchModule(const Args &) { return fail("not built yet"); }
int result2profileModule(const Args &) { return fail("not built yet"); }
int subtractdbsModule(const Args &) { return fail("not built yet"); }
int mergedbsModule(const Args &) { return fail("not built yet"); }
}
