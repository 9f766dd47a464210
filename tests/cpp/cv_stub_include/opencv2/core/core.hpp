// reached by `#include <opencv2/core/core.hpp>` under -Itests/cpp/cv_stub_include: the syntax stand-in, not OpenCV (see the file it includes)
#pragma once
#include "../../../cv_syntax_stub.h"
