// oracle/_stub/vbem — TEST INFRASTRUCTURE: included by CollapsedGibbsSampler.cpp, nothing of it is used there (the draws are std::gamma_distribution)
#pragma once
