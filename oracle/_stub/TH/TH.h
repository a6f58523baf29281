// intentionally empty: stands in for <TH/TH.h> (removed from modern torch); the reference file uses no TH symbol
