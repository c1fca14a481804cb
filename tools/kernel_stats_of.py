import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if (sys.argv[2] if len(sys.argv) > 2 else 'nn_ring') in r['Name']:
        print(r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
